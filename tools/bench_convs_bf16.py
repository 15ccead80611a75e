#!/usr/bin/env python3
"""Per-shape timing of the bf16 channels-last conv on the MRF shapes of BASELINE.json configs[4] (batch 64):
achieved HBM GB/s (algorithmic bytes: x + out [+ res + add], 2 bytes per element) and matrix TF/s.  Measurement tool."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from openvoice_amd.bf16 import PackedConvBf16, launch_conv_bf16  # noqa: E402
from bench_convs import timed  # noqa: E402


def pair_table(args):
    from openvoice_amd.bf16 import launch_pair_bf16, pair_bf16_supported
    dev, B = "cuda:0", args.batch
    print(f"B={B}: fused bf16 pair vs c1 + c2 launches; GB/s = algorithmic bytes of the FUSED form (x + out) / time")
    print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'pair ms':>8} {'GB/s':>7} {'c1+c2 ms':>9}")
    for c, L in [(64, 110208), (32, 220416)]:
        x = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
        t, out = torch.empty_like(x), torch.empty_like(x)
        for k in (3, 7, 11):
            for d in (1, 5):
                if not pair_bf16_supported(c, k, d):
                    continue
                c1 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=d)
                c2 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)

                def two():
                    launch_conv_bf16(c1, x, t, in_slope=0.1)
                    launch_conv_bf16(c2, t, out, in_slope=0.1, res=x)
                ms = timed(lambda: launch_pair_bf16(c1, c2, x, out), args.reps, 100.0)
                ms2 = timed(two, args.reps, 100.0)
                print(f"{c:>4} {L:>7} {k:>2} {d:>1} {ms:8.3f} {2 * 2.0 * B * c * L / ms / 1e6:7.0f} {ms2:9.3f}", flush=True)
        del x, t, out
    print("done")


def pair2_table(args):
    """Second-generation fused pair (activated tensors, C = 64 / 128) against the two launches of the round-3 path."""
    import torch.nn.functional as F
    from openvoice_amd.bf16 import launch_pair2_bf16, launch_pair_bf16, pair_bf16_supported
    dev, B = "cuda:0", args.batch
    print(f"B={B}: pair2 (ov_resblock_pair2_bf16cl) vs c1 + c2 launches [vs first-generation pair]; TF/s = both convs")
    print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'pair2 ms':>9} {'TF/s':>7} {'GB/s':>7} {'+add ms':>8} {'c1+c2 ms':>9} {'pair1 ms':>9}")
    for c, L in [(128, 55104), (64, 110208), (32, 220416)]:
        x = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
        xa = F.leaky_relu(x.float(), 0.1).to(torch.bfloat16)
        t, out, add = torch.empty_like(x), torch.empty_like(x), torch.randn_like(x)
        for k in (3, 7, 11):
            for d in (1, 5):
                c1 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=d)
                c2 = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=1)

                def two():
                    launch_conv_bf16(c1, x, t, in_slope=0.1, out_slope=0.1)
                    launch_conv_bf16(c2, t, out, in_slope=1.0, res=x)
                ms = timed(lambda: launch_pair2_bf16(c1, c2, xa, out, out_slope=0.1), args.reps, 100.0)
                msa = timed(lambda: launch_pair2_bf16(c1, c2, xa, out, add=add, scale=1.0 / 3.0), args.reps, 100.0)
                ms2 = timed(two, args.reps, 100.0)
                ms1 = (timed(lambda: launch_pair_bf16(c1, c2, x, out), args.reps, 100.0)
                       if pair_bf16_supported(c, k, d) else float("nan"))
                tf = 2 * 2.0 * c * c * k * L * B / ms / 1e9
                print(f"{c:>4} {L:>7} {k:>2} {d:>1} {ms:9.3f} {tf:7.1f} {2 * 2.0 * B * c * L / ms / 1e6:7.0f} {msa:8.3f} "
                      f"{ms2:9.3f} {ms1:9.3f}", flush=True)
        del x, xa, t, out, add
    print("done")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layouts", type=int, nargs="+", default=[0])
    ap.add_argument("--pair", action="store_true", help="time the fused ResBlock pair against its two launches only")
    ap.add_argument("--pair2", action="store_true", help="time the second-generation fused pair (C = 64 / 128)")
    args = ap.parse_args()
    if args.pair2:
        return pair2_table(args)
    if args.pair:
        return pair_table(args)
    dev, B = "cuda:0", args.batch
    print(f"B={B}  (HBM peak 8000 GB/s spec, ~6300 achievable; bf16 MFMA peak 2500 TF/s)")
    print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'epi':>8} {'ms':>8} {'GB/s':>8} {'TF/s':>7}")
    for c, L in [(256, 6888), (128, 55104), (64, 110208), (32, 220416)]:
        x = torch.randn(B, L, c, device=dev).to(torch.bfloat16)
        res, add = torch.randn_like(x), torch.randn_like(x)
        out = torch.empty_like(x)
        for k in (3, 7, 11):
            for d, mode in ((1, "plain"), (5, "plain"), (1, "res+add")):
                layer = PackedConvBf16(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, dil=d)
                kw = dict(in_slope=0.1)
                passes = 2
                if mode != "plain":
                    kw.update(res=res, add=add, scale=1.0 / 3.0)
                    passes = 4
                for lay in args.layouts:
                    if lay not in (0, 7, 8) and c <= 64:
                        continue
                    kw["layout"] = lay
                    ms = timed(lambda: launch_conv_bf16(layer, x, out, **kw), args.reps, 100.0)   # clocks ramped
                    gbs = passes * 2.0 * B * c * L / ms / 1e6
                    tf = 2.0 * c * c * k * L * B / ms / 1e9
                    print(f"{c:>4} {L:>7} {k:>2} {d:>1} {mode:>8} lay{lay} {ms:8.3f} {gbs:8.0f} {tf:7.1f}", flush=True)
        del x, res, add, out
    print("done")


if __name__ == "__main__":
    main()
