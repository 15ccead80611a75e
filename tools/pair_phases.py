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

# ---- bf16 channels-last pair (ov_resblock_pair_bf16cl) ----------------------------------------------------------
from openvoice_amd.bf16 import PackedConvBf16, launch_pair_bf16, pair_bf16_supported  # noqa: E402

PH16 = ["acc init", "chunk wait", "c1 k-steps", "t -> LDS", "t barrier", "c2 k-steps", "residual + add", "epilogue",
        "tail+barrier"]
B = 64
for c, L, TT in ((32, 220416, 256), (64, 110208, 128)):
    x = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    for k in (3, 7, 11):
        if not pair_bf16_supported(c, k, 1):
            continue
        c1 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)
        c2 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)
        for _ in range(20):
            launch_pair_bf16(c1, c2, x, out)
        dbg = torch.zeros(1024 * 12 * 9, dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_pair_bf16(c1, c2, x, out, dbg=dbg)
        e1.record()
        torch.cuda.synchronize()
        full = dbg.view(1024, 12, 9).double()
        full = full[full.sum(dim=(1, 2)) > 0]
        t, lt = full[:, :8], full[:, 8:, :3]
        tot = t.sum(-1).mean().item()
        nsteps = (L + (k - 1) // 2 + TT - 1) // TT * B / t.shape[0]
        print(f"bf16 C={c} k={k}: {e0.elapsed_time(e1):.3f} ms, {t.shape[0]} workgroups, {nsteps:.1f} steps each, "
              f"{tot / nsteps:.0f} ticks per step per wave")
        for q, name in enumerate(PH16):
            v = t[:, :, q].mean().item()
            print(f"    {name:15s} {100 * v / tot:5.1f} %   {v / nsteps:9.0f} ticks/step")
        print("    loader waves: " + ", ".join(f"{n} {lt[:, :, q].mean().item() / nsteps:.0f}" for q, n in
                                             enumerate(("barriers", "write", "issue"))) + " ticks/step")
