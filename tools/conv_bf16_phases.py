#!/usr/bin/env python3
"""Where does a matrix wave of the bf16 channels-last conv spend its time?  One launch per shape with the kernel's
phase timers on (ov_conv1d_bf16_params.dbg).  Measurement tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.bf16 import PackedConvBf16, launch_conv_bf16  # noqa: E402

PH = ["set-up", "chunk barriers", "k-step loops", "identity rounds + barrier A", "tile -> LDS", "barrier B", "global stores"]
dev, B = "cuda:0", 64
for c, L in ((128, 55104), (256, 6888), (64, 110208), (32, 220416)):
    x = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
    res = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    for k in (3, 11):
        layer = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)
        for mode in ("plain", "res"):
            kw = dict(in_slope=0.1)
            if mode == "res":
                kw.update(res=res)
            for _ in range(20):
                launch_conv_bf16(layer, x, out, **kw)
            dbg = torch.zeros(60000 * 32, dtype=torch.int64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch_conv_bf16(layer, x, out, dbg=dbg, **kw)
            e1.record()
            torch.cuda.synchronize()
            t = dbg.view(-1, 4, 8).double()
            t = t[t.sum(dim=(1, 2)) > 0]
            tiles = t[:, :, 7].mean().item()
            tot = t[:, :, :7].sum(-1).mean().item() / tiles
            print(f"C={c} k={k} {mode}: {e0.elapsed_time(e1):.3f} ms, {t.shape[0]} workgroups x {tiles:.1f} tiles, "
                  f"{tot:.0f} ticks per tile per wave")
            print("    " + "  ".join(f"{n} {t[:, :, q].mean().item() / tiles:.0f}" for q, n in enumerate(PH)))
