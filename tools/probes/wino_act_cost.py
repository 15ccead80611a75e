"""Probe (measurement only): what do the helpers' leaky-ReLU instructions cost?  Times the Winograd conv with in_slope = 0.1 (12 VALU
per staged item: a first conv of a pair) against in_slope = 1.0 (none: a second conv reading an activated tensor), plain form.
python tools/probes/wino_act_cost.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openvoice_amd import wino  # noqa: E402
from tools.bench_split3 import ramp, timed  # noqa: E402

DEV = "cuda:0"
gen = torch.Generator().manual_seed(3)
ramp()
for C, L in ((128, 55104), (256, 6888), (64, 110208), (32, 220416)):
    for K in (3, 7, 11):
        if not wino.supported(C, C, K, 1):
            continue
        x = torch.randn(32, C, L, generator=gen).to(DEV)
        out = torch.empty_like(x)
        layer = wino.PackedConvWino(torch.randn(C, C, K, generator=gen) * (C * K) ** -0.5, torch.zeros(C), DEV)
        row = {"C": C, "K": K}
        for slope in (0.1, 1.0):
            run = lambda: wino.launch_conv_wino(layer, x, C * L, out, C * L, 32, L, in_slope=slope)
            run()
            torch.cuda.synchronize()
            row[f"ms_in_slope_{slope}"] = round(min(timed(run, 10) for _ in range(3)), 4)
        print(json.dumps(row), flush=True)
        del x, out
