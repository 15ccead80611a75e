#!/usr/bin/env python3
"""What the reference API call costs end to end: ``ToneColorConverter.convert(path, src_se, tgt_se)`` (reference:
openvoice/api.py:141-160) on a 10 s file -- read + decode (+ resample), H2D, spectrogram, conversion, D2H, return -- per
stage and in total, fp32 path, opt-in split-precision path and captured-graph replay.  Files: 22.05 kHz 16-bit WAV (no
resampling), 44.1 kHz WAV (the kaiser_best resampler runs), and an MPEG-1 Layer III stream from the test fixtures (decoded by
openvoice_amd/mp3.py, 1.3 s of audio).  Measurement tool.
    python tools/bench_convert_file.py [--runs 20]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    args = ap.parse_args()
    from openvoice_amd import api, audio_io
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import default_converter_hparams
    hps = default_converter_hparams("v2")
    tmp = tempfile.mkdtemp()
    cfg = {"_version_": "v2", "data": dict(hps.data.items()), "model": dict(hps.model.items())}
    with open(os.path.join(tmp, "config.json"), "w") as fh:
        json.dump(cfg, fh)
    torch.save({"model": synthetic_state_dict(dict(hps.model.items()), 513, seed=1234)}, os.path.join(tmp, "checkpoint.pth"))
    tcc = api.ToneColorConverter(os.path.join(tmp, "config.json"), device="cuda:0", enable_watermark=False)
    tcc.load_ckpt(os.path.join(tmp, "checkpoint.pth"))
    gen = torch.Generator().manual_seed(1)
    se = [(0.1 * torch.randn(1, 256, 1, generator=gen)).to("cuda:0") for _ in range(2)]
    files = {}
    for sr in (22050, 44100):
        t = np.arange(10 * sr) / sr
        wave = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1370 * t + 1.0) +
                0.01 * np.random.default_rng(0).standard_normal(len(t))).astype(np.float32)
        files[f"wav_{sr}_10s"] = os.path.join(tmp, f"src{sr}.wav")
        audio_io.write(files[f"wav_{sr}_10s"], wave, sr)
    fixture = os.path.join(REPO, "tests", "golden", "mp3_syn_mpeg1_44100_mono_reservoir.npz")
    if os.path.exists(fixture):
        files["mp3_44100_1.3s"] = os.path.join(tmp, "src.mp3")
        with open(files["mp3_44100_1.3s"], "wb") as fh:
            fh.write(bytes(np.load(fixture)["stream"]))

    def timed(fn, runs):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(runs):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / runs * 1e3

    for name, path in files.items():
        rec = {"file": name}
        audio, _ = audio_io.load(path, sr=hps.data.sampling_rate)
        rec["seconds"] = round(len(audio) / hps.data.sampling_rate, 2)
        rec["load_ms"] = round(timed(lambda: audio_io.load(path, sr=hps.data.sampling_rate), max(3, args.runs // 4)), 3)
        y = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).unsqueeze(0)
        for mode in ("fp32", "split_bf16x3", "fp32_graph"):
            tcc.enable_split_bf16x3(mode == "split_bf16x3")
            tcc.use_graphs = mode == "fp32_graph"
            rec[f"device_ms_{mode}"] = round(timed(lambda: tcc.convert_batch(y, se[0], se[1], tau=0.3)[0].cpu(), args.runs), 3)
            rec[f"convert_ms_{mode}"] = round(timed(lambda: tcc.convert(path, se[0], se[1], tau=0.3), args.runs), 3)
        tcc.enable_split_bf16x3(False)
        tcc.use_graphs = False
        rec["real_time_factor_fp32"] = round(rec["seconds"] / (rec["convert_ms_fp32"] * 1e-3), 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
