#!/usr/bin/env python3
"""Where does a matrix wave of the fused ResBlock pair spend its time?  Runs one launch with the kernel's phase timers
on (ov_respair_params.dbg) and prints the mean shader-clock share of each phase.  Measurement tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.engine import PackedConv, launch_pair, pair_supported  # noqa: E402

PH = ["residual issue", "chunk wait", "c1 k-steps", "h -> LDS", "h barrier", "c2 k-steps", "epilogue", "tail+barrier"]
dev, B = "cuda:0", 32
for c, L in ((32, 220416), (64, 110208)):
    x = torch.randn(B, c, L, device=dev)
    out = torch.empty_like(x)
    for k in (3, 7, 11):
        if not pair_supported(c, k, 1):
            continue
        c1 = PackedConv(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, K=k, dil=1)
        c2 = PackedConv(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, K=k, dil=1)
        for _ in range(30):
            launch_pair(c1, c2, x, c * L, out, c * L, B, L)
        dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_pair(c1, c2, x, c * L, out, c * L, B, L, dbg=dbg)
        e1.record()
        torch.cuda.synchronize()
        t = dbg.view(512, 4, 8).double()
        t = t[t.sum(dim=(1, 2)) > 0]
        tot = t.sum(-1).mean().item()
        nsteps = (L + (k - 1) // 2 + (256 if c == 32 else 128) - 1) // (256 if c == 32 else 128) * B / t.shape[0]
        mf = 2 * (c // 2) * k * (c // 32) * (2 if c == 32 else 1) * 64   # MFMA cycles per wave per step (c1 + c2)
        print(f"C={c} k={k}: {e0.elapsed_time(e1):.3f} ms, {t.shape[0]} workgroups, {nsteps:.1f} steps each, "
              f"{tot / nsteps:.0f} ticks per step per wave (MFMA issue alone: {mf}); ticks are 100 MHz if the constant "
              f"clock, shader cycles otherwise")
        for q, name in enumerate(PH):
            v = t[:, :, q].mean().item()
            print(f"    {name:15s} {100 * v / tot:5.1f} %   {v / nsteps:9.0f} ticks/step")
