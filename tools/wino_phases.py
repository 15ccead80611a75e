#!/usr/bin/env python3
"""Phase timers of the Winograd-domain conv kernel (ov_conv1d_wino_params.dbg): shader-clock ticks per chunk for the
matrix waves (k-step loops, chunk barriers, epilogue + item set-up) and the helper waves (staging issue, transform,
raw write, barrier wait), averaged over all workgroups.  reference: openvoice/modules.py:296-309."""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from openvoice_amd import wino  # noqa: E402
from tools.bench_split3 import ramp, timed  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=128)
    ap.add_argument("--K", type=int, default=11)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--L", type=int, default=0, help="columns (default: the generator stage of C at 861 frames)")
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--frags", type=int, default=0)
    args = ap.parse_args()
    C, K, B = args.C, args.K, args.batch
    L = args.L or 861 * {256: 8, 128: 64, 64: 128, 32: 256}[C]
    gen = torch.Generator().manual_seed(0)
    w = torch.randn(C, C, K, generator=gen) * (C * K) ** -0.5
    layer = wino.PackedConvWino(w, torch.zeros(C), DEV, dil=args.dil)
    x = torch.randn(B, C, L, generator=gen).to(DEV)
    out = torch.empty_like(x)
    res = torch.randn(B, C, L, generator=gen).to(DEV) if args.res else None
    nwg = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    dbg = torch.zeros(nwg + 8, 8, 8, dtype=torch.int64, device=DEV)
    kw = dict(in_slope=0.1, res=res, res_bs=C * L if args.res else 0, frags=args.frags)
    nf = args.frags or 2
    nh = 2 * nf
    run = lambda: wino.launch_conv_wino(layer, x, C * L, out, C * L, B, L, **kw)
    ramp()
    ms = timed(run, 10)
    wino.launch_conv_wino(layer, x, C * L, out, C * L, B, L, dbg=dbg, **kw)
    torch.cuda.synchronize()
    d = dbg.double()
    live = d[:, 0, 7] > 0
    d = d[live]
    chunks = d[:, :, 7:8].clamp_min(1)
    per = (d / chunks).mean(0)             # [6 waves][8]
    G = (K + 2) // 3
    from openvoice_amd import _lib
    ci = _lib.call("ov_conv1d_wino_chunk", K, C)
    mfma_cycles = ci * G // 2 * 6 * 64 * nf
    rec = {"tool": "wino_phases", "C": C, "K": K, "dil": args.dil, "B": B, "L": L, "res": args.res, "frags": nf, "ms": ms,
           "mfma_issue_cycles_per_chunk": mfma_cycles, "workgroups": int(live.sum().item()),
           "matrix_ticks_per_chunk": {k: round(per[:4, i].mean().item()) for i, k in enumerate(["k_loops", "barrier", "epilogue", "item_setup"])},
           "helper_ticks_per_chunk": {k: round(per[4:4 + nh, i].mean().item()) for i, k in enumerate(["stage_issue", "transform", "raw_write", "barrier"])}}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
