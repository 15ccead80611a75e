#!/usr/bin/env python3
"""CPU baseline, reference vs port, on the SAME host (build container only: needs /root/reference).

    python tools/cpu_reference_vs_port.py > profiles/rNN_cpu_reference_vs_port_container.json

bench.py times the unmodified reference when it can import it and the oracle ("port") otherwise -- the GPU box has no
/root/reference.  This records both on one machine, with the B = 32 run forced, so that the port figure reported from
the GPU box can be read as a stand-in for the reference figure ('weak' item 8).  Measurement tool.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openvoice_amd.params import synthetic_state_dict  # noqa: E402
from openvoice_amd.utils import default_converter_hparams  # noqa: E402

hps = default_converter_hparams("v2")
cfg = dict(hps.model.items())
sd = synthetic_state_dict(cfg, 513, seed=1234)
out = {}
for kind, want in (("reference", True), ("port", False)):
    out[kind] = bench.cpu_baseline(sd, cfg, 10.0, budget_s=60.0, want_reference=want)
ref, port = out["reference"], out["port"]
out["port_over_reference"] = {
    "b1_all_threads": round(port["value"] / ref["value"], 3),
    "b1_one_thread": round(port["one_thread"]["value"] / ref["one_thread"]["value"], 3),
    "b32_all_threads": (round(port["batch32"]["value"] / ref["batch32"]["value"], 3)
                        if port.get("batch32") and ref.get("batch32") else None)}
json.dump(out, sys.stdout, indent=1)
print()
