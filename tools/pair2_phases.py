#!/usr/bin/env python3
"""Where does a matrix wave of the second-generation fused bf16 pair (ov_resblock_pair2_bf16cl) spend its time?  One
launch per shape with the kernel's phase timers on (ov_respair2_bf16_params.dbg): mean shader-clock ticks per step and
phase, next to the bare MFMA issue time of the step.  Measurement tool.   python tools/pair2_phases.py [C ...]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.bf16 import PackedConvBf16, launch_pair2_bf16  # noqa: E402

PH = ["barrier A", "c1 (+cells)", "t -> LDS", "barrier B", "c2 k-steps", "barrier C", "epilogue"]
dev, B = "cuda:0", 64
shapes = {128: (55104, 128), 64: (110208, 256), 32: (220416, 512)}
for c in ([int(a) for a in sys.argv[1:]] or [128, 64]):
    L, TT = shapes[c]
    xa = F.leaky_relu(torch.randn(B, L, c, device=dev), 0.1).to(torch.bfloat16)
    out = torch.empty_like(xa)
    cases = ((3, 1, 0), (7, 1, 0), (11, 1, 0), (11, 5, 0), (3, 1, 1), (11, 1, 1), (3, 1, 2), (11, 1, 2))
    if os.environ.get("P2_CASES"):        # "k,d,flags;k,d,flags"
        cases = tuple(tuple(int(v) for v in c_.split(",")) for c_ in os.environ["P2_CASES"].split(";"))
    for k, d, fl in cases:
        if fl == 2 and c != 128:
            continue
        c1 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=d)
        c2 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)
        for _ in range(30):
            launch_pair2_bf16(c1, c2, xa, out, out_slope=0.1, exp_flags=fl)
        dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_pair2_bf16(c1, c2, xa, out, out_slope=0.1, dbg=dbg, exp_flags=fl)
        e1.record()
        torch.cuda.synchronize()
        full = dbg.view(256, 8, 8).double()
        full = full[full[:, :4, 7].sum(dim=1) > 0]
        t, lt = full[:, :4], full[:, 4:]
        steps = t[:, :, 7].mean().item()
        tot = t[:, :, :7].sum(-1).mean().item()
        mf = 2 * (c // 16) * k * 4 * 32      # MFMA issue cycles per wave per step: 2 convs x (C/16 k-blocks x K) x 4 x 32
        ms = e0.elapsed_time(e1)
        print(f"C={c} k={k} d={d} flags={fl}: {ms:.3f} ms, {t.shape[0]} workgroups x {steps:.1f} pseudo-steps, "
              f"{tot / steps:.0f} ticks per step per wave (bare MFMA issue {mf}) -> clock {tot / ms / 1e6:.2f} GHz, "
              f"pipe busy {mf * steps / tot:.3f}")
        for q, name in enumerate(PH):
            v = t[:, :, q].mean().item()
            per_wave = " ".join(f"{t[:, w, q].mean().item() / steps:7.0f}" for w in range(4))
            print(f"    {name:12s} {100 * v / tot:5.1f} %   {v / steps:9.0f} ticks/step   per wave: {per_wave}")
        print("    loader waves: " + ", ".join(f"{n} {lt[:, :, q].mean().item() / steps:.0f}" for q, n in enumerate(
            ("wait DMA/stores", "barrier A", "tile->regs", "DMA issue", "barrier B", "barrier C", "stores"))) + " ticks/step")
