"""Generate tests/golden/*.pt by running the UNMODIFIED reference here (ORACLE tooling).

Runs only in the build container, where /root/reference exists:  python oracle/make_golden.py
The GPU box has no /root/reference; tests there read the committed fixtures.

What is recorded, per case: the seeded inputs (waveform, speaker embeddings, posterior noise,
lengths) and the reference outputs -- ``spectrogram_torch`` (openvoice/mel_processing.py:40-75),
``SynthesizerTrn.voice_conversion`` (openvoice/models.py:492-499; returns o_hat, y_mask,
(z, z_p, z_hat)) and ``SynthesizerTrn.ref_enc`` (openvoice/models.py:339-359).  Weights are the
calibrated synthetic set ``openvoice_amd.params.synthetic_state_dict(seed)``; a fingerprint of
them is stored so drift of the generator is detected rather than silently compared.
The reference's only RNG draw, ``torch.randn_like`` at models.py:220, is patched to return the
recorded noise tensor.
"""
import os
import sys
import types
import warnings

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = "/root/reference"

from openvoice_amd.params import stress_state_dict, synthetic_state_dict, synthetic_tts_state_dict  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG  # noqa: E402

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED = 1234


def import_reference():
    sys.path.insert(0, REFERENCE)
    for name in ("librosa", "librosa.filters"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    sys.modules["librosa.filters"].mel = lambda *a, **k: None
    from openvoice import models as ref_models
    from openvoice.mel_processing import spectrogram_torch
    return ref_models, spectrogram_torch


def synth_wave(batch, samples, seed):
    """Sum of 5 random sinusoids (80-4000 Hz) + 0.01 N(0,1), peak 0.9."""
    gen = torch.Generator().manual_seed(seed)
    t = torch.arange(samples, dtype=torch.float64) / 22050.0
    freqs = 80.0 + (4000.0 - 80.0) * torch.rand(batch, 5, generator=gen, dtype=torch.float64)
    phases = 2 * torch.pi * torch.rand(batch, 5, generator=gen, dtype=torch.float64)
    amps = 0.2 + torch.rand(batch, 5, generator=gen, dtype=torch.float64)
    wave = (amps[..., None] * torch.sin(2 * torch.pi * freqs[..., None] * t + phases[..., None])).sum(1)
    wave = wave + 0.01 * torch.randn(batch, samples, generator=gen, dtype=torch.float64)
    wave = 0.9 * wave / wave.abs().amax(dim=1, keepdim=True)
    return wave.float()


def weight_fingerprint(sd):
    keys = ["dec.ups.0.weight_g", "dec.resblocks.4.convs2.1.weight_v", "enc_q.enc.in_layers.7.weight_v",
            "flow.flows.4.post.weight", "ref_enc.gru.weight_hh_l0"]
    return {k: float(sd[k].double().abs().sum()) for k in keys}


CASES = [
    # name, batch, frames, lengths (None = full), zero_g, per-item speaker embeddings, tau
    dict(name="vc_b2_t17", batch=2, frames=17, lengths=None, zero_g=False, per_item_g=False, tau=0.3),
    dict(name="vc_b3_t65_ragged_zero_g", batch=3, frames=65, lengths=[65, 37, 20], zero_g=True,
         per_item_g=True, tau=0.3),
    dict(name="vc_b1_t40_tau0", batch=1, frames=40, lengths=None, zero_g=True, per_item_g=False, tau=0.0),
    # the benchmark frame count (10 s @ 22.05 kHz), V2 converter settings; compact=True stores only what the
    # test needs (waveform in, o_hat + latent checksums out) to keep the fixture small
    dict(name="vc_b1_t861_benchmark_length", batch=1, frames=861, lengths=None, zero_g=True, per_item_g=False,
         tau=0.3, compact=True),
    # high dynamic range: params.stress_state_dict(gain 4) -- latents and flow shifts 4x larger,
    # so the flow's forward / reverse cancellation and every accumulation run at 4x the magnitude; tau = 1 adds the
    # full posterior noise; ragged so the masks are exercised at that magnitude too
    dict(name="vc_b2_t64_stress_gain4", batch=2, frames=64, lengths=[64, 41], zero_g=True, per_item_g=True, tau=1.0,
         stress=4.0),
]


TTS_WEIGHT_SEED = 4321
TTS_VOCAB, TTS_SPEAKERS = 68, 10
TTS_CASES = [
    # name, token lengths (batch = len), speaker ids, noise_scale, length_scale, noise_scale_w, sdp_ratio
    dict(name="tts_b3_tx23_ragged", lengths=[23, 15, 7], sid=[0, 3, 9], noise_scale=0.667, length_scale=1.0,
         noise_scale_w=0.6, sdp_ratio=0.2),
    dict(name="tts_b1_tx40_slow", lengths=[40], sid=[5], noise_scale=0.667, length_scale=1.3,
         noise_scale_w=0.6, sdp_ratio=0.2),
    dict(name="tts_b2_tx5_tiny", lengths=[5, 3], sid=[1, 2], noise_scale=0.0, length_scale=1.0,
         noise_scale_w=0.8, sdp_ratio=0.5),
]


def make_tts_golden(ref_models):
    """BaseSpeakerTTS model half: SynthesizerTrn.infer (openvoice/models.py:467-490) and its parts.
    The two RNG draws -- torch.randn in the stochastic duration predictor (models.py:175) and
    torch.randn_like for the prior sample (models.py:487) -- are patched to return recorded tensors."""
    sd = synthetic_tts_state_dict(CONVERTER_MODEL_CONFIG, TTS_VOCAB, TTS_SPEAKERS, 513, seed=TTS_WEIGHT_SEED)
    model = ref_models.SynthesizerTrn(TTS_VOCAB, 513, n_speakers=TTS_SPEAKERS, **CONVERTER_MODEL_CONFIG).eval()
    model.load_state_dict(sd, strict=True)
    schema = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    torch.save(schema, os.path.join(GOLDEN_DIR, "tts_state_dict_schema.pt"))
    for case in TTS_CASES:
        lengths = torch.tensor(case["lengths"], dtype=torch.long)
        B, Tx = len(case["lengths"]), max(case["lengths"])
        gen = torch.Generator().manual_seed(300 + len(case["name"]))
        tokens = torch.randint(0, TTS_VOCAB, (B, Tx), generator=gen)
        sid = torch.tensor(case["sid"], dtype=torch.long)
        noise_w = torch.randn(B, 2, Tx, generator=gen)
        noise_z_full = torch.randn(B, 192, 64 * Tx, generator=gen)
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def fake_randn(*shape, **kw):
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert shape == (B, 2, Tx), shape
            return noise_w.clone()

        torch.randn = fake_randn
        torch.randn_like = lambda x, *a, **k: noise_z_full[:, :, :x.shape[2]].clone()
        try:
            with torch.no_grad():
                o, attn, y_mask, (z, z_p, m_p, logs_p) = model.infer(
                    tokens, lengths, sid=sid, noise_scale=case["noise_scale"], length_scale=case["length_scale"],
                    noise_scale_w=case["noise_scale_w"], sdp_ratio=case["sdp_ratio"])
                x, m_tok, logs_tok, x_mask = model.enc_p(tokens, lengths)
                g = model.emb_g(sid).unsqueeze(-1)
                logw_sdp = model.sdp(x, x_mask, g=g, reverse=True, noise_scale=case["noise_scale_w"])
                logw_dp = model.dp(x, x_mask, g=g)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        Ty = z.shape[2]
        rec = dict(case=case, weight_seed=TTS_WEIGHT_SEED, n_vocab=TTS_VOCAB, n_speakers=TTS_SPEAKERS,
                   tokens=tokens, lengths=lengths, sid=sid, noise_w=noise_w, noise_z=noise_z_full[:, :, :Ty].clone(),
                   x=x, m_tok=m_tok, logs_tok=logs_tok, x_mask=x_mask, logw_sdp=logw_sdp, logw_dp=logw_dp,
                   o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p)
        torch.save(rec, os.path.join(GOLDEN_DIR, case["name"] + ".pt"))
        w = attn[:, 0].sum(1)
        print(f"{case['name']}: Ty {Ty} o {tuple(o.shape)} |o|max {o.abs().max():.3f} durations/token "
              f"{w[0, :8].tolist()} z std {z.std():.3f}")


def main():
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, spectrogram_torch = import_reference()
    sd = synthetic_state_dict(CONVERTER_MODEL_CONFIG, 513, seed=WEIGHT_SEED)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if not only:
        make_tts_golden(ref_models)
    if "--tts-only" in sys.argv:
        return
    for case in CASES:
        if only and case["name"] not in only:
            continue
        model = ref_models.SynthesizerTrn(0, 513, n_speakers=0, zero_g=case["zero_g"],
                                          **CONVERTER_MODEL_CONFIG).eval()
        sd_case = stress_state_dict(sd, case["stress"]) if case.get("stress") else sd
        missing, unexpected = model.load_state_dict(sd_case, strict=True)
        b, t = case["batch"], case["frames"]
        seed = 100 + len(case["name"])
        wave = synth_wave(b, 256 * t, seed)
        gen = torch.Generator().manual_seed(seed + 1)
        gshape = (b if case["per_item_g"] else 1, 256, 1)
        g_src = 0.3 * torch.randn(gshape, generator=gen)
        g_tgt = 0.3 * torch.randn(gshape, generator=gen)
        noise = torch.randn(b, 192, t, generator=gen)
        lengths = torch.tensor(case["lengths"] or [t] * b, dtype=torch.long)
        with torch.no_grad():
            spec = spectrogram_torch(wave, 1024, 22050, 256, 1024, center=False)
            assert spec.shape == (b, 513, t), spec.shape
            real_randn_like = torch.randn_like
            torch.randn_like = lambda x, *a, **k: noise.to(x.dtype)
            try:
                o_hat, y_mask, (z, z_p, z_hat) = model.voice_conversion(
                    spec, lengths, sid_src=g_src, sid_tgt=g_tgt, tau=case["tau"])
            finally:
                torch.randn_like = real_randn_like
            se = model.ref_enc(spec.transpose(1, 2))
        if only and case["name"] not in only:
            continue
        rec = dict(case=case, weight_seed=WEIGHT_SEED, weight_fingerprint=weight_fingerprint(sd),
                   wave=wave, g_src=g_src, g_tgt=g_tgt, noise=noise, lengths=lengths, spec=spec,
                   o_hat=o_hat, y_mask=y_mask, z=z, z_p=z_p, z_hat=z_hat, ref_enc=se)
        if case.get("compact"):
            # inputs are regenerated from the seeds by the test (wave, noise, g); keep outputs in fp16-free form:
            # o_hat in full, the latents as per-channel sums (a checksum that still localises an error)
            rec = dict(case=case, weight_seed=WEIGHT_SEED, weight_fingerprint=weight_fingerprint(sd), seed=seed,
                       o_hat=o_hat, z_sum=z.sum(2), z_p_sum=z_p.sum(2), z_hat_sum=z_hat.sum(2), ref_enc=se,
                       spec_sum=spec.sum(2))
        path = os.path.join(GOLDEN_DIR, case["name"] + ".pt")
        torch.save(rec, path)
        print(f"{case['name']}: o_hat {tuple(o_hat.shape)} |o|max {o_hat.abs().max():.3f} "
              f"z std {z.std():.3f} |z_p-z|max {(z_p - z).abs().max():.3f} -> {path}")
    if only:
        return
    # Reference encoder on a longer clip (70 frames -> 2 GRU steps)
    model = ref_models.SynthesizerTrn(0, 513, n_speakers=0, **CONVERTER_MODEL_CONFIG).eval()
    model.load_state_dict(sd, strict=True)
    wave = synth_wave(2, 256 * 200, 77)
    with torch.no_grad():
        spec = spectrogram_torch(wave, 1024, 22050, 256, 1024, center=False)
        se = model.ref_enc(spec.transpose(1, 2))
    torch.save(dict(wave=wave, spec=spec, ref_enc=se, weight_seed=WEIGHT_SEED,
                    weight_fingerprint=weight_fingerprint(sd)),
               os.path.join(GOLDEN_DIR, "ref_enc_b2_t200.pt"))
    # Checkpoint schema of the reference model (names + shapes), for the loader tests.
    schema = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    torch.save(schema, os.path.join(GOLDEN_DIR, "converter_state_dict_schema.pt"))
    print("schema tensors:", len(schema))


if __name__ == "__main__":
    main()
