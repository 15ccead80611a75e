#!/usr/bin/env python3
"""Latency of one conversion at small batch, eager launches vs HIP-graph replay (engine.GraphedConversion).
Measurement tool: python tools/bench_latency.py [--batches 1 4 32] [--seconds 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4, 32])
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from openvoice_amd.engine import ConverterEngine
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    dev = "cuda:0"
    eng = ConverterEngine(synthetic_state_dict(CFG, 513, seed=1234), CFG, 513, dev, zero_g=True)
    T = int(args.seconds * 22050) // 256
    gen = torch.Generator().manual_seed(0)
    rows = []
    for B in args.batches:
        spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(dev)
        lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
        g_src, g_tgt = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(dev), (0.3 * torch.randn(1, 256, 1, generator=gen)).to(dev)
        noise = torch.randn(B, 192, T, generator=gen).to(dev)
        graphed = eng.graphed(B, T, 0.3)
        calls = {"eager": lambda: eng.voice_conversion(spec, lengths, g_src, g_tgt, tau=0.3, noise=noise),
                 "graph": lambda: graphed(spec, lengths, g_src, g_tgt, noise=noise)}
        rec = {"batch": B, "frames": T}
        for name, fn in calls.items():
            t_end = time.perf_counter() + 0.3          # pre-heat: clocks ramped before timing
            while time.perf_counter() < t_end:
                fn()
            torch.cuda.synchronize()
            steps = args.steps if B < 16 else max(3, args.steps // 5)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            rec[name + "_ms"] = round(ms, 3)
            rec[name + "_rtf"] = round(B * args.seconds / (ms * 1e-3), 1)
        rec["speedup"] = round(rec["eager_ms"] / rec["graph_ms"], 3)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    return rows


if __name__ == "__main__":
    main()
