#!/usr/bin/env python3
"""HBM traffic of the split-precision conv launches (ovks3::conv1d_split3_kernel) from two rocprofv3 --pmc passes of
`bench.py --split-bf16x3` (FETCH_SIZE, WRITE_SIZE), per kernel instance and per conversion, next to the ALGORITHMIC bytes:
every tensor element is three bf16 planes = 6 bytes; a launch reads its input once, its residual once (residual form) and
writes its output once -- B x L x C x 6 bytes each; the packed weights (C x C x K x 6 bytes, L2-resident) are not counted.

    python tools/pmc_traffic_split.py FETCH_DIR WRITE_DIR [STEPS_IN_PASS [BATCH [FRAMES]]] > profiles/rNN_split3_traffic.json

Units and corrections as in tools/pmc_traffic.py (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): counters in KiB,
FETCH_SIZE x 2 for wide streaming reads on gfx950, both factors also measured on three 1 GiB copies of the same pass.
Measurement tool."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import GIB, read  # noqa: E402

NAME = re.compile(r"conv1d_split3_kernel<(\d+), (\d+), (\d+), (\d+), (true|false|0|1), (\d+)>")
# released generator: channels -> columns per frame of the stage that runs at that width (upsample 8, 8, 2, 2)
COLS_PER_FRAME = {256: 8, 128: 64, 64: 128}


def collect(rows):
    by = {}
    for name, val, _ in rows:
        m = NAME.search(name)
        if m:
            k, d, cin, cot, res, npr = m.groups()
            by.setdefault((int(k), int(d), int(cin), res in ("true", "1"), int(npr)), []).append(val)
    copies = sorted((v for n, v, g in rows if "copyBuffer" in n or "copy_kernel" in n.lower()), reverse=True)[:3]
    return by, copies


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    frames = int(sys.argv[5]) if len(sys.argv) > 5 else 861
    f_by, f_cal = collect(read(fetch_dir, "FETCH_SIZE"))
    w_by, w_cal = collect(read(write_dir, "WRITE_SIZE"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openvoice_amd.hostinfo import split3_source_digest
    rf = GIB / (sum(f_cal) / len(f_cal) * 1024) if f_cal else 2.0
    wf = GIB / (sum(w_cal) / len(w_cal) * 1024) if w_cal else 1.0
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --split-bf16x3, counters in KiB",
           "split3_source_digest": split3_source_digest(), "batch": batch, "frames": frames, "steps_in_pass": steps,
           "read_factor": round(rf, 3), "write_factor": round(wf, 3),
           "factors": "measured on 3 x 1 GiB device-to-device copies of the same pass" if f_cal and w_cal else
                      "nominal (read x 2, write x 1)",
           "instances": []}
    tot_meas = tot_alg = 0.0
    launches = 0
    for key in sorted(f_by):
        if key not in w_by or len(f_by[key]) != len(w_by[key]):
            continue
        k, d, c, res, npr = key
        n = len(f_by[key])
        fetch = rf * sum(f_by[key]) / n * 1024
        write = wf * sum(w_by[key]) / n * 1024
        tensor = 6.0 * batch * frames * COLS_PER_FRAME.get(c, 0) * c
        alg_r, alg_w = tensor * (2 if res else 1), tensor
        out["instances"].append({"K": k, "dil": d, "C": c, "residual": res, "products": npr, "launches_per_step": n / steps,
                                 "read_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                                 "algorithmic_read": round(alg_r), "algorithmic_write": round(alg_w),
                                 "read_over_algorithmic": round(fetch / alg_r, 3) if alg_r else None,
                                 "write_over_algorithmic": round(write / alg_w, 3) if alg_w else None})
        tot_meas += (fetch + write) * n / steps
        tot_alg += (alg_r + alg_w) * n / steps
        launches += n
    out["launches_per_step"] = launches / steps
    out["bytes_per_step"] = round(tot_meas)
    out["algorithmic_bytes_per_step"] = round(tot_alg)
    out["traffic_over_algorithmic"] = round(tot_meas / tot_alg, 3) if tot_alg else None
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
