#!/usr/bin/env python3
"""Gate table of the Winograd-domain fp32 conv: ONE shape first -- generator stage 1, C = 128,
k = 11, d = 1, batch 32 x 55 104 columns -- against the direct fp32 MFMA conv on the same tensors.

  (a) time: ov_conv1d_wino_f32 vs ov_conv1d_f32 (conv1d_mfma_kernel<11,1,...>), alternating launches, clocks ramped;
      gate >= 1.35x;
  (b) error: max-abs vs a float64 conv of the same fp32 operands, for both kernels, on the calibrated AND the gain-4
      stress inputs; gate: Winograd <= 16 x direct.

Prints a table and one JSON line; --shapes adds the other (C, K) with an instance; --res times the residual form too.
reference: openvoice/modules.py:296-309."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from openvoice_amd import wino  # noqa: E402
from openvoice_amd.engine import LRELU_SLOPE, PackedConv, launch_conv  # noqa: E402
from openvoice_amd.params import effective_weight, synthetic_state_dict  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG  # noqa: E402
from tools.bench_split3 import f64_conv, ramp, timed  # noqa: E402

DEV = "cuda:0"
MACS_PER_4 = wino.PRODUCTS_PER_TILE      # executed multiply-accumulates per 4 outputs and (co, ci)


def one_shape(sd, C, K, d, B, L, reps, err_items=2, with_res=False, frags=0):
    rb = {256: 0, 128: 3, 64: 6, 32: 9}[C] + {3: 0, 7: 1, 11: 2}[K]
    n = {1: 0, 3: 1, 5: 2}[d]
    gen = torch.Generator().manual_seed(C + K + d)
    x0 = torch.randn(B, C, L, generator=gen).to(DEV)
    w = effective_weight(sd, f"dec.resblocks.{rb}.convs1.{n}").float()
    b = sd[f"dec.resblocks.{rb}.convs1.{n}.bias"].float()
    direct = PackedConv(w, b, DEV, K=K, dil=d)
    wn = wino.PackedConvWino(w, b, DEV, dil=d)
    row = dict(C=C, K=K, dil=d, B=B, L=L, gflop=2.0 * C * C * K * L * B / 1e9,
               gflop_executed=2.0 * C * C * MACS_PER_4[K] / 4.0 * L * B / 1e9)
    out_d, out_w = torch.empty(B, C, L, device=DEV), torch.empty(B, C, L, device=DEV)
    for name in ("calibrated", "stress_gain4"):
        x = x0 if name == "calibrated" else 4.0 * x0 * torch.exp2(2.0 * torch.randn(B, C, 1, generator=gen).to(DEV))
        run_d = lambda: launch_conv(direct, x, 0, C * L, out_d, 0, C * L, B, L, in_slope=LRELU_SLOPE)
        run_w = lambda: wino.launch_conv_wino(wn, x, C * L, out_w, C * L, B, L, in_slope=LRELU_SLOPE, frags=frags)
        out_w.fill_(float("nan"))
        run_d()
        run_w()
        torch.cuda.synchronize()
        # both against float64 on the first items, and against each other everywhere (a stale column would show as NaN)
        ref = f64_conv(F.leaky_relu(x[:err_items], LRELU_SLOPE), w.to(DEV), b.to(DEV), K, d)
        e_d = (out_d[:err_items].double() - ref).abs().max().item()
        e_w = (out_w[:err_items].double() - ref).abs().max().item()
        rec = dict(out_absmax=ref.abs().max().item(), err_direct=e_d, err_wino=e_w, err_ratio=e_w / max(e_d, 1e-30),
                   max_abs_wino_vs_direct_all_items=(out_w - out_d).abs().max().item())
        if name == "calibrated":
            ramp()
            td, tw = [], []
            for _ in range(3):
                td.append(timed(run_d, reps))
                tw.append(timed(run_w, reps))
            rec.update(ms_direct=min(td), ms_wino=min(tw), speedup=min(td) / min(tw),
                       tflops_direct=row["gflop"] / min(td), tflops_wino_executed=row["gflop_executed"] / min(tw),
                       tflops_wino_algorithmic_equivalent=row["gflop"] / min(tw))
            if with_res:
                res = torch.randn(B, C, L, generator=gen).to(DEV)
                run_dr = lambda: launch_conv(direct, x, 0, C * L, out_d, 0, C * L, B, L, in_slope=LRELU_SLOPE, res=res,
                                             res_bs=C * L)
                run_wr = lambda: wino.launch_conv_wino(wn, x, C * L, out_w, C * L, B, L, in_slope=LRELU_SLOPE, res=res,
                                                       res_bs=C * L, frags=frags)
                run_dr(); run_wr()
                rec["res_form_max_abs_wino_vs_direct"] = (out_w - out_d).abs().max().item()
                td, tw = [], []
                for _ in range(3):
                    td.append(timed(run_dr, reps))
                    tw.append(timed(run_wr, reps))
                rec.update(ms_direct_res=min(td), ms_wino_res=min(tw), speedup_res=min(td) / min(tw))
        row[name] = rec
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=861)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", action="store_true", help="every (C, K, dilation) with an instance, not only the gate shape")
    ap.add_argument("--res", action="store_true", help="also time the residual form")
    ap.add_argument("--frags", type=int, default=0, help="128-column fragments per matrix wave (0 = dispatcher)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sd = synthetic_state_dict(CONVERTER_MODEL_CONFIG, 513, seed=1234)
    shapes = [(128, 11, 1)]
    if args.shapes:
        shapes += [(C, K, d) for C in (128, 256, 64, 32) for K in (3, 7, 11) for d in (1, 3, 5)
                   if (C, K, d) != (128, 11, 1) and wino.supported(C, C, K, d)]
    rows = []
    for C, K, d in shapes:
        L = args.frames * {256: 8, 128: 64, 64: 128, 32: 256}[C]
        row = one_shape(sd, C, K, d, args.batch, L, args.reps, with_res=args.res, frags=args.frags)
        rows.append(row)
        c, s = row["calibrated"], row["stress_gain4"]
        print(f"C={C} K={K} d={d}: direct {c['ms_direct']:.3f} ms ({c['tflops_direct']:.1f} TF/s)  wino {c['ms_wino']:.3f} ms "
              f"({c['tflops_wino_executed']:.1f} TF/s executed, {c['tflops_wino_algorithmic_equivalent']:.1f} algorithmic-equivalent)"
              f"  speedup {c['speedup']:.2f}x | err vs f64: direct {c['err_direct']:.2e} wino {c['err_wino']:.2e} "
              f"(x{c['err_ratio']:.1f}); stress: direct {s['err_direct']:.2e} wino {s['err_wino']:.2e} (x{s['err_ratio']:.1f})"
              + (f" | res form {c['ms_direct_res']:.3f} vs {c['ms_wino_res']:.3f} ms ({c['speedup_res']:.2f}x)" if args.res else ""),
              flush=True)
    g = rows[0]
    gate = {"speedup_ok": g["calibrated"]["speedup"] >= 1.35,
            "error_ok": max(g["calibrated"]["err_ratio"], g["stress_gain4"]["err_ratio"]) <= 16.0}
    gate["passed"] = gate["speedup_ok"] and gate["error_ok"]
    line = {"tool": "bench_wino", "device": torch.cuda.get_device_name(0), "gate_shape": "C=128 K=11 d=1", "gate": gate,
            "rows": rows}
    print(json.dumps(line))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(line, f, indent=1)


if __name__ == "__main__":
    main()
