#!/usr/bin/env python3
"""Determinism soak of the small-batch paths: N conversions of the same batch-1 / batch-2 input with a fixed noise tensor,
every output compared bit for bit with the first (a race in the WaveNet row-split launch pair, or anywhere else, would show
as a differing run), fp32 path and opt-in split-precision path.  Measurement / verification tool.
    python tools/soak_small_batch.py [--runs 300]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=300)
    args = ap.parse_args()
    from bench import SAMPLE_RATE, synth_wave
    from openvoice_amd.mel_processing import spectrogram_torch
    from openvoice_amd.models import SynthesizerTrn
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import default_converter_hparams
    dev = torch.device("cuda:0")
    hps = default_converter_hparams("v2")
    cfg = dict(hps.model.items())
    model = SynthesizerTrn(0, 513, n_speakers=0, **cfg)
    model.load_state_dict(synthetic_state_dict(cfg, 513, seed=1234), strict=True)
    model = model.to(dev).eval()
    gen = torch.Generator().manual_seed(1)
    se = [(0.1 * torch.randn(1, 256, 1, generator=gen)).to(dev) for _ in range(2)]
    d = hps.data
    eng = model.engine()
    for B in (1, 2):
        wave = synth_wave(B, 10 * SAMPLE_RATE, 1000 + B, dev)
        spec = spectrogram_torch(wave, d.filter_length, d.sampling_rate, d.hop_length, d.win_length, center=False)
        lengths = torch.full((B,), spec.shape[2], dtype=torch.int64, device=dev)
        noise = torch.randn(B, 192, spec.shape[2], generator=torch.Generator().manual_seed(B)).to(dev)
        for split in (False, True):
            eng.use_split_bf16x3(split)
            first = None
            differing = 0
            for _ in range(args.runs):
                o = model.voice_conversion(spec, lengths, se[0], se[1], tau=0.3, noise=noise)[0]
                if first is None:
                    first = o.clone()
                elif not torch.equal(o, first):
                    differing += 1
            torch.cuda.synchronize()
            print(json.dumps({"batch": B, "split_bf16x3": split, "runs": args.runs, "runs_differing_from_the_first": differing,
                              "finite": bool(torch.isfinite(first).all())}), flush=True)
        eng.use_split_bf16x3(False)


if __name__ == "__main__":
    main()
