#!/usr/bin/env python3
"""Throughput of the V1 base-speaker TTS path (BASELINE.json configs[3]: SynthesizerTrn.infer, batch 16, one
MI355X) with the calibrated synthetic weights.  Measurement tool, not the bench.py contract line.

    python tools/bench_tts.py [--batch 16] [--tokens 100] [--steps 5] [--cpu]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tokens", type=int, default=100, help="symbol ids per utterance (blanks included)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on one utterance")
    ap.add_argument("--full-padding", action="store_true",
                    help="compute the generator on the whole padded batch, as the reference does (round-2 figure); "
                         "default: length-aware work lists (infer(skip_padding=True): length + 16 frames per "
                         "utterance, valid samples bit-identical -- tests/test_gpu_limits.py)")
    ap.add_argument("--split-bf16x3", type=int, default=0, choices=(0, 6, 3),
                    help="opt-in: the generator's MRF stages on the split-precision kernels with 6 or 3 plane products "
                         "(ConverterEngine.use_split_bf16x3; those stages compute the padded batch: no length-aware lists)")
    args = ap.parse_args()
    from openvoice_amd.models import SynthesizerTrn
    from openvoice_amd.params import synthetic_tts_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    dev = "cuda:0"
    sd = synthetic_tts_state_dict(CFG)
    model = SynthesizerTrn(68, 513, n_speakers=10, **CFG)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    if args.split_bf16x3:
        model.engine().core.use_split_bf16x3(True, products=args.split_bf16x3)
    gen = torch.Generator().manual_seed(0)
    B, Tx = args.batch, args.tokens
    tokens = torch.randint(0, 68, (B, Tx), generator=gen).to(dev)
    lengths = torch.full((B,), Tx, dtype=torch.long, device=dev)
    sid = (torch.arange(B) % 10).to(dev)
    noise_w = torch.randn(B, 2, Tx, generator=gen).to(dev)
    noise_z = torch.randn(B, 192, 16 * Tx, generator=gen).to(dev)

    def timed(skip):
        def step():
            return model.infer(tokens, lengths, sid=sid, noise_scale=0.667, noise_scale_w=0.6, length_scale=1.0,
                               noise_w=noise_w, noise_z=noise_z, skip_padding=skip)
        for _ in range(args.warmup):
            o, _, y_mask, _ = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o, _, y_mask, _ = step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, o, y_mask

    dt_full, o_full, y_mask = timed(False)
    dt, o, y_mask = (dt_full, o_full, y_mask) if args.full_padding else timed(True)
    frames = int(y_mask.sum())
    audio_s = frames * 256 / 22050.0
    # Algorithmic FLOPs (2 x MACs of the convs; SURVEY.md section 8d's per-frame hand count): the generator and the
    # reverse flow run at FRAME rate on the padded batch (B x max frames), the text encoder / duration predictors at
    # token rate (enc_p 4.4, sdp 0.33, dp 0.21 GFLOP per ~100-symbol utterance).
    Ty = int(y_mask.shape[2])
    per_frame = 614.8e6 + 4 * 3.54e6            # generator + 4 coupling layers (one flow direction)
    token_rate = (4.4e9 + 0.33e9 + 0.21e9) * B * Tx / 100.0
    flops_padded = per_frame * B * Ty + token_rate
    # with length-aware work lists the generator computes min(Ty, length + 16) frames per utterance (flow: all)
    per_utt = y_mask[:, 0].sum(1)
    gen_frames = float(torch.clamp(per_utt + 16, max=Ty).sum()) if not args.full_padding else float(B * Ty)
    flops = 614.8e6 * gen_frames + 4 * 3.54e6 * B * Ty + token_rate
    valid_equal = None
    if not args.full_padding:
        valid_equal = all(torch.equal(o[b, :, :256 * int(n)], o_full[b, :, :256 * int(n)]) for b, n in enumerate(per_utt))
    out = {"workload": f"SynthesizerTrn.infer, batch {B} x {Tx} symbols, fp32, synthetic weights"
                       + ("" if args.full_padding else ", length-aware generator work lists (skip_padding)"),
           "ms_per_batch": round(dt * 1e3, 3), "utterances_per_s": round(B / dt, 2),
           "ms_per_batch_full_padding": round(dt_full * 1e3, 3),
           "valid_samples_bit_identical_to_full_padding": valid_equal,
           "generator_frames_computed": int(gen_frames), "generator_frames_padded_batch": B * Ty,
           "audio_s_per_batch": round(audio_s, 2), "real_time_factor": round(audio_s / dt, 1),
           "frames_per_utterance": round(frames / B, 1), "padded_frames": Ty,
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                        "frac": round(flops / dt / 157.3e12, 4), "alg_tflop_per_batch": round(flops / 1e12, 3),
                        "alg_tflop_per_padded_batch": round(flops_padded / 1e12, 3),
                        "note": "whole infer() incl. its host sync (Ty = y_lengths.max(), as in the reference, "
                                "models.py:478-480) and the token-rate kernels; FLOPs counted on the frames the "
                                "generator actually computes (length + 16 per utterance with skip_padding; the "
                                "whole padded batch with --full-padding, as the reference's unmasked generator)"}}
    if args.cpu:
        from oracle import tts_oracle
        from openvoice_amd.hostinfo import usable_cpus
        cores = usable_cpus(32)
        torch.set_num_threads(cores)
        a = lambda t: t[:1].cpu()
        with torch.no_grad():
            tts_oracle.infer(sd, CFG, a(tokens), a(lengths), a(sid), a(noise_w), a(noise_z), 0.667, 1.0, 0.6)
            t0 = time.perf_counter()
            o_c = tts_oracle.infer(sd, CFG, a(tokens), a(lengths), a(sid), a(noise_w), a(noise_z), 0.667, 1.0, 0.6)[0]
            dtc = time.perf_counter() - t0
        out["cpu_oracle"] = {"cores": cores, "s_per_utterance": round(dtc, 3),
                             "real_time_factor": round(o_c.shape[2] / 22050.0 / dtc, 2),
                             # the generator is unmasked: in a padded batch the last ~13 frames of an item see
                             # conv_pre(0) + bias instead of the signal edge, so compare away from the tail
                             "max_abs_vs_gpu_item0_excl_last_20_frames":
                                 float((o[0, 0, :o_c.shape[2] - 5120].cpu() - o_c[0, 0, :-5120]).abs().max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
