#!/usr/bin/env python3
"""BASELINE.json configs[4]: HiFi-GAN generator with bf16 activations, batch 64, 10 s utterances (T = 861), one
MI355X -- time per batch, utterances/s, algorithmic HBM GB/s (every tensor pass of the launch sequence, 2 bytes per
element) against the 8 TB/s spec / ~6.3 TB/s achievable, next to the fp32 generator on the same input.
Measurement tool.   python tools/bench_decoder_bf16.py [--batch 64] [--steps 5]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def alg_bytes(cfg, B, T, esize):
    from openvoice_amd.bf16 import generator_alg_bytes
    return generator_alg_bytes(cfg, B, T, esize)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=861)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--chain-streams", type=int, default=None,
                    help="HIP streams for the independent ResBlock chains of a stage (default: GeneratorBf16's, 3; "
                         "1 = the serial one-stream order, which is what the rocprofv3 --pmc passes see anyway: "
                         "counter collection serialises kernels)")
    ap.add_argument("--pmc-calibration", action="store_true",
                    help="after the timed region, three 1 GiB device-to-device copies (a known byte count) so that a "
                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass over this command can be calibrated")
    ap.add_argument("--counters", default=None,
                    help="JSON written by tools/bf16_counters.py from rocprofv3 --pmc passes over THIS command: its "
                         "measured HBM traffic / matrix-busy / clock are reported next to the algorithmic figures")
    args = ap.parse_args()
    from openvoice_amd.bf16 import GeneratorBf16
    from openvoice_amd.engine import ConverterEngine
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    dev, B, T = "cuda:0", args.batch, args.frames
    sd = synthetic_state_dict(CFG, 513, seed=1234)
    gen = torch.Generator().manual_seed(0)
    z = torch.randn(B, 192, T, generator=gen).to(dev)
    g = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(dev)
    dec = GeneratorBf16(sd, CFG, dev)
    if args.chain_streams is not None:
        dec.chain_streams = args.chain_streams

    def timeit(fn):
        for _ in range(2):
            o = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, o

    dt, o16 = timeit(lambda: dec.decode(z, g))
    dt_serial = None
    if dec.chain_streams > 1:          # the same generator on one stream, same process: what the concurrency buys
        keep, dec.chain_streams = dec.chain_streams, 1
        dt_serial, o_serial = timeit(lambda: dec.decode(z, g))
        dec.chain_streams = keep
        assert torch.equal(o_serial, o16), "concurrent chains must not change the result"
    out = {"workload": f"HiFi-GAN generator, bf16 activations (channels-last), batch {B} x {T} frames (10 s), one MI355X",
           "ms_per_batch": round(dt * 1e3, 3), "utterances_per_s": round(B / dt, 1),
           "chain_streams": dec.chain_streams,
           "ms_per_batch_one_stream": round(dt_serial * 1e3, 3) if dt_serial else None,
           "real_time_factor": round(B * T * 256 / 22050.0 / dt, 1),
           "alg_hbm_GB": round(alg_bytes(CFG, B, T, 2) / 1e9, 2),
           "alg_hbm_GBps": round(alg_bytes(CFG, B, T, 2) / dt / 1e9, 1), "hbm_peak_GBps": 8000, "hbm_achievable_GBps": 6300,
           "frac_of_hbm_peak": round(alg_bytes(CFG, B, T, 2) / dt / 8e12, 3),
           "alg_tflops": round(529.44e9 * B * T / 861 / dt / 1e12, 1), "bf16_mfma_peak_tflops": 2500,
           # what the bf16 pipe sustains on random operands under the power limit (tools/micro/mfma_bf16_ceiling.hip,
           # profiles/r03_s2_mfma_bf16_power_limited_ceiling.txt: 1.66-1.73 PFLOP/s over three boxes)
           "bf16_mfma_sustained_tflops_measured": 1700,
           "frac_of_sustained_mfma": round(529.44e9 * B * T / 861 / dt / 1700e12, 3)}
    out["decode_passes_in_process"] = 2 + args.steps
    if args.counters and os.path.exists(args.counters):
        with open(args.counters) as fh:
            c = json.load(fh)
        out["traffic_hbm_GB"] = c.get("traffic_GB_per_pass")
        out["traffic_over_algorithmic"] = (round(c["traffic_GB_per_pass"] / out["alg_hbm_GB"], 3)
                                           if c.get("traffic_GB_per_pass") else None)
        out["traffic_hbm_GBps"] = round(c["traffic_GB_per_pass"] / dt, 1) if c.get("traffic_GB_per_pass") else None
        out["mfma_busy"] = c.get("mfma_busy")
        out["shader_clock_ghz"] = c.get("shader_clock_ghz")
        out["counters_source"] = c.get("source")
    if args.pmc_calibration:
        src = torch.zeros(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        del src, dst
    if not args.no_fp32:
        eng = ConverterEngine(sd, CFG, 513, dev, zero_g=False)
        cond = eng._linear(g.reshape(1, -1), eng.dec_cond_w, eng.dec_cond_b)
        dt32, o32 = timeit(lambda: eng.decode(z, cond))
        err = (o16 - o32).abs()
        out["fp32_generator_ms_per_batch"] = round(dt32 * 1e3, 3)
        out["speedup_vs_fp32"] = round(dt32 / dt, 2)
        out["max_abs_vs_fp32"] = round(float(err.max()), 4)
        out["rel_rms_vs_fp32"] = round(float(err.pow(2).mean().sqrt() / o32.pow(2).mean().sqrt()), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
