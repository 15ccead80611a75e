#!/usr/bin/env python3
"""Where does a matrix wave of the MFMA conv kernel spend its time?  Needs the OV_EXP=3 measurement build
(scripts/conv_phases.sh), whose matrix waves write phase timers through ov_conv1d_params.out2.  Measurement tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402

PH = ["set-up + preload", "chunk barriers", "k-step loops", "bookkeeping + epilogue"]
dev, B = "cuda:0", 32
shapes = [(128, 55104), (256, 6888), (64, 110208)]
for c, L in shapes:
    x = torch.randn(B, c, L, device=dev)
    res = torch.randn(B, c, L, device=dev)
    out = torch.empty_like(x)
    for k in (3, 11):
        layer = PackedConv(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, K=k, dil=1)
        for mode in ("plain", "res"):
            kw = dict(in_slope=0.1)
            if mode == "res":
                kw.update(res=res, res_bs=c * L)
            for _ in range(30):
                launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, **kw)
            dbg = torch.zeros(16384 * 4 * 8, dtype=torch.int64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, out2=dbg.view(torch.float32), **kw)
            e1.record()
            torch.cuda.synchronize()
            t = dbg.view(-1, 4, 8).double()
            t = t[t[:, :, 4].sum(dim=1) > 0]
            tiles = t[:, :, 4].mean().item()
            tot = t[:, :, :4].sum(-1).mean().item()
            mf = (c // 2) * k * 4 * 64    # MFMA issue cycles per wave per tile (4 per k-step, 64 cycles each)
            print(f"C={c} k={k} {mode}: {e0.elapsed_time(e1):.3f} ms, {t.shape[0]} workgroups x {tiles:.1f} tiles, "
                  f"{tot / tiles:.0f} ticks per tile per wave (its own MFMA issue: {mf}, x2 when the SIMD is shared)")
            for q, name in enumerate(PH):
                v = t[:, :, q].mean().item()
                print(f"    {name:24s} {100 * v / tot:5.1f} %   {v / tiles:9.0f} ticks/tile")
