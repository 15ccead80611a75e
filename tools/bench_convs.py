#!/usr/bin/env python3
"""Per-shape timing of the MFMA conv kernel on the MRF (ResBlock) shapes of the benchmark batch
(B = 32, 10 s utterances).  Measurement tool, not part of the product path.

    python tools/bench_convs.py [--tpw 0 1 2 4] [--reps 5]
    OV_CONV_IMPL=v1 python tools/bench_convs.py      # round-1 single-role kernel, for A/B
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402

PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tpw", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--kernels", type=int, nargs="+", default=[3, 7, 11])
    args = ap.parse_args()
    dev = "cuda:0"
    B = args.batch
    stages = [(256, 6888), (128, 55104), (64, 110208), (32, 220416)]
    print(f"impl={os.environ.get('OV_CONV_IMPL', 'v2')} B={B}")
    print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'epi':>8} {'tpw':>3} {'ms':>8} {'TF/s':>7} {'%peak':>6}")
    for c, L in stages:
        x = torch.randn(B, c, L, device=dev)
        res = torch.randn(B, c, L, device=dev)
        add = torch.randn(B, c, L, device=dev)
        out = torch.empty(B, c, L, device=dev)
        for k in args.kernels:
            for d, mode in ((1, "plain"), (5, "plain"), (1, "res+add")):
                w = torch.randn(c, c, k) * (c * k) ** -0.5
                layer = PackedConv(w, torch.zeros(c), dev, K=k, dil=d)
                for tpw in args.tpw:
                    if tpw and os.environ.get("OV_CONV_IMPL") == "v1":
                        continue
                    kw = dict(in_slope=0.1, tiles_per_wg=tpw)
                    if mode != "plain":
                        kw.update(res=res, res_bs=c * L, add=add, add_bs=c * L, scale=1.0 / 3.0)
                    for _ in range(2):
                        launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / args.reps
                    tf = 2.0 * c * c * k * L * B / ms / 1e9
                    print(f"{c:>4} {L:>7} {k:>2} {d:>1} {mode:>8} {tpw:>3} {ms:8.3f} {tf:7.1f} {100 * tf / PEAK:6.1f}")
        del x, res, add, out
    print("done")


if __name__ == "__main__":
    main()
