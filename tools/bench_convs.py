#!/usr/bin/env python3
"""Per-shape timing of the MFMA conv kernel on the MRF (ResBlock) shapes of the benchmark batch
(B = 32, 10 s utterances).  Measurement tool, not part of the product path.

    python tools/bench_convs.py [--tpw 0 2] [--loaders 0 1 2 4] [--tiles 0 3 4] [--reps 5]

--tpw: 0 = the dispatcher's launch rule, n > 0 = n tiles per workgroup, -1 = force the persistent launch.
tile ids: 0 = dispatcher's choice, 1..4 = 128x128, 64x256, 32x512, 32x256; a (tile, loaders)
combination that is not instantiated for a shape is skipped.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402

PEAK = 157.3


def timed(fn, reps, warm_ms):
    """Average ms per call of fn, measured with the clocks already ramped: the GPU is kept busy for ~warm_ms
    right before (and with no host sync between warm-up and) the timed launches.  With two warm-up launches
    only -- what this tool did until profiles/r01_s36 -- the first shapes after every host-side weight pack ran
    10-20 % below their steady-state rate (62 % vs 79 % of peak on k = 3, C = 128), an artefact of the
    power-state ramp, not of the kernel."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    n = max(2, int(warm_ms / max(e0.elapsed_time(e1), 1e-3)))
    for _ in range(n):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tpw", type=int, nargs="+", default=[0])
    ap.add_argument("--loaders", type=int, nargs="+", default=[0])
    ap.add_argument("--tiles", type=int, nargs="+", default=[0])
    ap.add_argument("--chunks", type=int, nargs="+", default=[0])
    ap.add_argument("--channels", type=int, nargs="+", default=[256, 128, 64, 32])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--warm-ms", type=float, default=100.0, help="GPU-busy time before each timed region")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--kernels", type=int, nargs="+", default=[3, 7, 11])
    ap.add_argument("--modes", nargs="+", default=["plain1", "plain5", "res", "res+add"],
                    help="plain1 / plain5 = conv1 with dilation 1 / 5, res = conv2 with the residual (16 of the 18 conv2 "
                         "launches of a stage), res+add = conv2 with residual and MRF running sum (the other 2)")
    ap.add_argument("--pair", action="store_true", help="time the fused ResBlock pair against its two-launch equivalent")
    ap.add_argument("--nwg", type=int, nargs="+", default=[0], help="--pair: workgroup counts (0 = one per resident slot)")
    ap.add_argument("--skip-single", action="store_true", help="skip the single-conv table")
    ap.add_argument("--wn", action="store_true", help="also time the WaveNet k5 gate conv and the 1x1 res/skip conv")
    args = ap.parse_args()
    dev = "cuda:0"
    B = args.batch
    stages = [(256, 6888), (128, 55104), (64, 110208), (32, 220416)]
    from openvoice_amd._lib import OvError
    print(f"B={B}")
    print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'epi':>8} {'tile':>4} {'nld':>3} {'ch':>2} {'tpw':>3} {'ms':>8} {'TF/s':>7} {'%peak':>6}")
    for c, L in stages:
        if c not in args.channels or args.skip_single:
            continue
        x = torch.randn(B, c, L, device=dev)
        res = torch.randn(B, c, L, device=dev)
        add = torch.randn(B, c, L, device=dev)
        out = torch.empty(B, c, L, device=dev)
        for k in args.kernels:
            for d, mode in ((1, "plain"), (5, "plain"), (1, "res"), (1, "res+add")):
                if (mode + str(d) if mode == "plain" else mode) not in args.modes:
                    continue
                w = torch.randn(c, c, k) * (c * k) ** -0.5
                layer = PackedConv(w, torch.zeros(c), dev, K=k, dil=d)
                import itertools
                for tile, nld, tpw, chunk in itertools.product(args.tiles, args.loaders, args.tpw, args.chunks):
                    kw = dict(in_slope=0.1, tiles_per_wg=tpw, tile=tile, loaders=nld, chunk=chunk)
                    if mode == "res":
                        kw.update(res=res, res_bs=c * L)
                    elif mode == "res+add":
                        kw.update(res=res, res_bs=c * L, add=add, add_bs=c * L, scale=1.0 / 3.0)
                    try:
                        ms = timed(lambda: launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, **kw), args.reps,
                                   args.warm_ms)
                    except OvError:
                        continue
                    tf = 2.0 * c * c * k * L * B / ms / 1e9
                    print(f"{c:>4} {L:>7} {k:>2} {d:>1} {mode:>8} {tile:>4} {nld:>3} {chunk:>2} {tpw:>3} {ms:8.3f} {tf:7.1f} "
                          f"{100 * tf / PEAK:6.1f}", flush=True)
        del x, res, add, out
    if args.pair:
        import itertools
        from openvoice_amd.engine import launch_pair, pair_supported
        print(f"{'C':>4} {'L':>7} {'k':>2} {'d':>1} {'pair ms':>8} {'TF/s':>7} {'%peak':>6} {'c1+c2 ms':>9} {'%peak':>6}")
        for c, L in stages:
            if c not in args.channels:
                continue
            x = torch.randn(B, c, L, device=dev)
            t = torch.empty(B, c, L, device=dev)
            out = torch.empty(B, c, L, device=dev)
            for k in args.kernels:
                for d in (1, 5):
                    if not pair_supported(c, k, d):
                        continue
                    c1 = PackedConv(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, K=k, dil=d)
                    c2 = PackedConv(torch.randn(c, c, k) * (c * k) ** -0.5, torch.zeros(c), dev, K=k, dil=1)

                    def two():
                        launch_conv(c1, x, 0, c * L, t, 0, c * L, B, L, in_slope=0.1)
                        launch_conv(c2, t, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=x, res_bs=c * L)
                    for nwg in args.nwg:
                        ms = timed(lambda: launch_pair(c1, c2, x, c * L, out, c * L, B, L, nwg=nwg), args.reps, args.warm_ms)
                        ms2 = timed(two, args.reps, args.warm_ms)
                        fl = 2 * 2.0 * c * c * k * L * B
                        print(f"{c:>4} {L:>7} {k:>2} {d:>1} {ms:8.3f} {fl / ms / 1e9:7.1f} {100 * fl / ms / 1e9 / PEAK:6.1f} "
                              f"{ms2:9.3f} {100 * fl / ms2 / 1e9 / PEAK:6.1f}  nwg={nwg}", flush=True)
            del x, t, out
    if args.wn:
        from openvoice_amd._lib import EPI_GATE, EPI_RESSKIP
        from openvoice_amd.engine import gate_row_order, padded_frames
        H, T = 192, 861
        for ld in (T, padded_frames(T)):
            x = torch.randn(B, H, ld, device=dev)
            acts = torch.empty(B, H, ld, device=dev)
            skip = torch.zeros(B, H, ld, device=dev)
            mask = torch.ones(B, ld, device=dev)
            cond = torch.randn(B, 2 * H, device=dev)
            order = gate_row_order(H)
            l_in = PackedConv((torch.randn(2 * H, H, 5) * (5 * H) ** -0.5)[order], torch.zeros(2 * H), dev, K=5, cout=H)
            l_rs = PackedConv(torch.randn(2 * H, H, 1) * H ** -0.5, torch.zeros(2 * H), dev, K=1)
            for name, fl, fn in (
                    ("wn_in k5 gate", 2.0 * 2 * H * H * 5 * T * B,
                     lambda n: launch_conv(l_in, x, 0, H * ld, acts, 0, H * ld, B, T, epi=EPI_GATE, bias_b=cond,
                                           bias_b_bs=2 * H, rows=2 * H, x_ld=ld, out_ld=ld, loaders=n)),
                    ("wn_rs 1x1", 2.0 * 2 * H * H * T * B,
                     lambda n: launch_conv(l_rs, acts, 0, H * ld, x, 0, H * ld, B, T, epi=EPI_RESSKIP, out2=skip,
                                           out2_bs=H * ld, mask=mask, mask_bs=ld, split=H, x_ld=ld, out_ld=ld,
                                           loaders=n))):
                for nld in args.loaders:
                    try:
                        ms = timed(lambda: fn(nld), 10, args.warm_ms)
                    except OvError:
                        continue
                    print(f"{name:>14} ld={ld} nld={nld} {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF/s "
                          f"{100 * fl / ms / 1e9 / PEAK:6.1f} %", flush=True)
        # the same layer as ONE launch (ov_wn_layer_f32), per tile width
        from openvoice_amd.engine import launch_wn_layer, wn_fused_row_order, wn_pack
        ld = padded_frames(T)
        x, out = torch.randn(B, H, ld, device=dev), torch.empty(B, H, ld, device=dev)
        skip, mask, cond = torch.zeros(B, H, ld, device=dev), torch.ones(B, ld, device=dev), torch.randn(B, 2 * H, device=dev)
        fo = wn_fused_row_order(H)
        layer = dict(hidden=H, K=5, w_in=wn_pack((torch.randn(2 * H, H, 5) * (5 * H) ** -0.5)[fo], dev),
                     b_in=torch.zeros(2 * H, device=dev), w_rs=wn_pack(torch.randn(2 * H, H, 1) * H ** -0.5, dev),
                     b_rs=torch.zeros(2 * H, device=dev))
        fl = 2.0 * 2 * H * H * 6 * T * B
        for width in (0, 64, 96, 112, 128):
            ms = timed(lambda: launch_wn_layer(layer, x, out, skip, mask, B, T, ld, cond=cond, cond_bs=2 * H,
                                               width=width, mask_bs=ld), 10, args.warm_ms)
            print(f"fused wn layer ld={ld} width={width or 'auto'} {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF/s "
                  f"{100 * fl / ms / 1e9 / PEAK:6.1f} %", flush=True)
    print("done")


if __name__ == "__main__":
    main()
