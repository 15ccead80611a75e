#!/usr/bin/env python3
"""BASELINE.json configs[4], per stage: where the bf16 generator's batch-64 pass goes and how far each part is from the
ceiling that binds it.  Every launch of ``GeneratorBf16.decode`` is bracketed with HIP events on
the launch stream and priced with its algorithmic FLOPs (non-zero taps) and HBM bytes (every tensor pass, 2 bytes per
element); a part's ceiling is the LARGER of flops / 1.7 PFLOP/s -- the bf16 matrix rate this chip sustains under its
power limit (profiles/r04_s30: 2.5 PF is the 2.4 GHz figure) -- and bytes / 6.3 TB/s (the achievable HBM stream rate).
Prints a table and one JSON line.  reference: openvoice/models.py:272-291."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_CEIL_PF, HBM_CEIL_TB = 1.7, 6.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=861)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from openvoice_amd import bf16 as m
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    dev = "cuda:0"
    sd = synthetic_state_dict(CFG, 513, seed=1234)
    gen = m.GeneratorBf16(sd, CFG, dev)
    B, T = args.batch, args.frames
    z = torch.randn(B, CFG["inter_channels"], T, device=dev)
    g = torch.randn(1, CFG["gin_channels"], 1, device=dev)
    stage_of = {CFG["upsample_initial_channel"] >> (i + 1): i for i in range(len(CFG["upsample_rates"]))}
    rec = []
    orig_launch, orig_pair2, orig_pair = m._launch, m.launch_pair2_bf16, m.launch_pair_bf16

    def timed(tag, flops, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        rec.append((tag, flops, nbytes, e0, e1))

    def _launch(layer, x, out, L, in_slope=1.0, scale=1.0, res=None, add=None, phase_s=0, **kw):
        Bx = x.shape[0]
        if phase_s:       # ConvTranspose as a 3-tap phase conv: two non-zero taps per phase
            co = layer.cout // phase_s
            tag = (stage_of[co], "ups")
            flops = 2.0 * layer.cin * co * 2 * phase_s * L * Bx
            nbytes = 2.0 * Bx * L * (layer.cin + layer.cout)
        elif layer.cin not in stage_of and layer.cout not in stage_of or layer.cin != layer.cout:
            tag = (-1, "conv_pre")
            flops = 2.0 * layer.cin * layer.cout * layer.K * L * Bx
            nbytes = 2.0 * Bx * L * (layer.cin + layer.cout)
        else:
            tag = (stage_of[layer.cout], f"mrf k={layer.K} (single convs)")
            flops = 2.0 * layer.cin * layer.cout * layer.K * L * Bx
            nbytes = 2.0 * Bx * L * layer.cout * (2 + (res is not None) + (add is not None))
        timed(tag, flops, nbytes, lambda: orig_launch(layer, x, out, L, in_slope=in_slope, scale=scale, res=res, add=add,
                                                      phase_s=phase_s, **kw))

    def pair_any(orig, kind):
        def f(c1, c2, x, out, add=None, **kw):
            Bx, L, C = x.shape
            flops = 2 * 2.0 * C * C * c1.K * L * Bx
            nbytes = 2.0 * Bx * L * C * (2 + (add is not None))
            timed((stage_of[C], f"mrf k={c1.K} ({kind})"), flops, nbytes, lambda: orig(c1, c2, x, out, add=add, **kw))
        return f

    m._launch, m.launch_pair2_bf16, m.launch_pair_bf16 = _launch, pair_any(orig_pair2, "fused pair2"), pair_any(orig_pair, "fused pair")
    for _ in range(2):
        gen.decode(z, g)
    rec.clear()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        gen.decode(z, g)
    e1.record()
    torch.cuda.synchronize()
    whole = e0.elapsed_time(e1) / args.steps
    rows = {}
    for tag, fl, nb, a, b in rec:
        r = rows.setdefault(tag, [0.0, 0.0, 0.0, 0])
        r[0] += a.elapsed_time(b) / args.steps; r[1] += fl / args.steps; r[2] += nb / args.steps; r[3] += 1
    out_rows, tot = [], 0.0
    print(f"{'stage':>5} {'part':<28} {'n':>3} {'ms':>7} {'PF/s':>6} {'TB/s':>6} {'of 1.7 PF':>9} {'of 6.3 TB/s':>11} {'binding':>8} {'ceiling ms':>10}")
    for tag in sorted(rows):
        ms, fl, nb, n = rows[tag]
        n //= args.steps
        pf, tb = fl / ms / 1e12, nb / ms / 1e9
        ceil_ms = max(fl / (MFMA_CEIL_PF * 1e12), nb / (HBM_CEIL_TB * 1e9))
        bind = "mfma" if fl / (MFMA_CEIL_PF * 1e12) >= nb / (HBM_CEIL_TB * 1e9) else "hbm"
        tot += ms
        out_rows.append(dict(stage=tag[0], part=tag[1], launches=n, ms=round(ms, 3), pflops=round(pf, 3), tb_per_s=round(tb, 3),
                             frac_mfma=round(pf / MFMA_CEIL_PF, 3), frac_hbm=round(tb / HBM_CEIL_TB, 3), binding=bind,
                             ceiling_ms=round(ceil_ms, 3), frac_of_binding_ceiling=round(ceil_ms / ms, 3)))
        print(f"{tag[0]:>5} {tag[1]:<28} {n:>3} {ms:>7.3f} {pf:>6.3f} {tb:>6.3f} {pf / MFMA_CEIL_PF:>9.3f} {tb / HBM_CEIL_TB:>11.3f} {bind:>8} {ceil_ms:>10.3f}")
    ceil_total = sum(r["ceiling_ms"] for r in out_rows)
    print(f"sum of bracketed launches {tot:.2f} ms; whole pass (events, no per-launch brackets would be ~0.1 ms less) {whole:.2f} ms; "
          f"sum of per-part ceilings {ceil_total:.2f} ms")
    line = dict(tool="bf16_stage_table", batch=B, frames=T, ms_per_pass=round(whole, 3), bracketed_ms=round(tot, 3),
                ceilings=dict(mfma_pflops=MFMA_CEIL_PF, hbm_tb_per_s=HBM_CEIL_TB), sum_of_ceilings_ms=round(ceil_total, 3), rows=out_rows)
    print(json.dumps(line))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(line, f, indent=1)


if __name__ == "__main__":
    main()
