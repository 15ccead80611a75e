"""Probe (measurement only): one vs two 128-column fragments per matrix wave (NF = 1: two workgroups per CU cover each
other's epilogues, twice the weight stream; NF = 2: one workgroup per CU) on the dilation-1 shapes with 128-row M-blocks,
plain and residual form.   python tools/probes/wino_frags_ab.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openvoice_amd.params import synthetic_state_dict  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG  # noqa: E402
from tools.bench_wino import one_shape  # noqa: E402

sd = synthetic_state_dict(CONVERTER_MODEL_CONFIG, 513, seed=1234)
for C in (128, 256):
    for K in (3, 7, 11):
        L = 861 * {256: 8, 128: 64}[C]
        row = {"C": C, "K": K}
        for frags in (1, 2):
            r = one_shape(sd, C, K, 1, 32, L, 10, with_res=True, frags=frags)["calibrated"]
            row[f"frags{frags}_ms"] = round(r["ms_wino"], 4)
            row[f"frags{frags}_res_ms"] = round(r["ms_wino_res"], 4)
        print(json.dumps(row), flush=True)
