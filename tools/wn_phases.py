#!/usr/bin/env python3
"""Where does a matrix wave of the fused WaveNet layer spend its time?  One launch with the kernel's phase timers on
(ov_wn_layer_params.dbg); prints the mean shader-clock share of each phase.  Measurement tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import launch_wn_layer, padded_frames, wn_fused_row_order, wn_pack  # noqa: E402

PH = ["first chunk wait", "gate-conv k-steps", "chunk waits", "gate", "operands + barrier", "res/skip k-steps", "epilogue"]
dev, H, T = "cuda:0", 192, 861
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
widths = [int(w) for w in sys.argv[2:]] or [0]
ld = padded_frames(T)
x, out = torch.randn(B, H, ld, device=dev), torch.empty(B, H, ld, device=dev)
skip, mask, cond = torch.zeros(B, H, ld, device=dev), torch.ones(B, ld, device=dev), torch.randn(B, 2 * H, device=dev)
layer = dict(hidden=H, K=5, w_in=wn_pack((torch.randn(2 * H, H, 5) * (5 * H) ** -0.5)[wn_fused_row_order(H)], dev),
             b_in=torch.zeros(2 * H, device=dev), w_rs=wn_pack(torch.randn(2 * H, H, 1) * H ** -0.5, dev),
             b_rs=torch.zeros(2 * H, device=dev))
for width in widths:
    w = _lib.call("ov_wn_layer_tile", B, T, width)
    nwg = B * ((T + w - 1) // w)
    for _ in range(300):
        launch_wn_layer(layer, x, out, skip, mask, B, T, ld, cond=cond, cond_bs=2 * H, width=width, mask_bs=ld)
    dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch_wn_layer(layer, x, out, skip, mask, B, T, ld, cond=cond, cond_bs=2 * H, width=width, mask_bs=ld, dbg=dbg)
    e1.record()
    torch.cuda.synchronize()
    t = dbg.view(nwg, 8, 8).double()
    ph, start = t[:, :, :7], t[:, :, 7]
    tot = ph.sum(-1).mean().item()
    mfma = (w // 16) * 3 * 32 * 2 * (240 + 48)      # both waves of a SIMD
    print(f"B={B} width={w}: {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups, {tot:.0f} ticks per wave "
          f"(MFMA issue of the SIMD's two waves: {mfma}); start spread {start.max().item() - start.min().item():.0f}, "
          f"end spread {(start + ph.sum(-1)).max().item() - start.min().item():.0f} ticks")
    for q, name in enumerate(PH):
        v = ph[:, :, q].mean().item()
        print(f"    {name:20s} {100 * v / tot:5.1f} %   {v:9.0f} ticks   (max over waves {ph[:, :, q].max().item():.0f})")
