"""Pins the oracle (oracle/vc_oracle.py) to outputs of the unmodified reference.

The fixtures under tests/golden/ were produced by oracle/make_golden.py, which imports the
reference from /root/reference.  Tolerances: the restatement folds weight-norm once instead of
per forward and otherwise issues the same ATen ops, so agreement is at fp32 round-off level.
"""
import glob
import os

import pytest
import torch

from oracle import vc_oracle
from oracle.make_golden import weight_fingerprint
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG

VC_CASES = ["vc_b2_t17", "vc_b3_t65_ragged_zero_g", "vc_b1_t40_tau0", "vc_b2_t64_stress_gain4"]


def case_state_dict(rec, synth_sd):
    """The weights a fixture was generated with: the calibrated synthetic set, or its high-dynamic-range variant."""
    from openvoice_amd.params import stress_state_dict
    gain = rec["case"].get("stress")
    return stress_state_dict(synth_sd, gain) if gain else synth_sd


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)


def test_fixtures_present(golden_dir):
    assert len(glob.glob(os.path.join(golden_dir, "*.pt"))) >= 5


@pytest.mark.parametrize("name", VC_CASES)
def test_weight_generator_has_not_drifted(golden_dir, synth_sd, name):
    rec = _load(golden_dir, name)
    now = weight_fingerprint(synth_sd)
    for key, val in rec["weight_fingerprint"].items():
        assert abs(now[key] - val) <= 1e-9 * abs(val), key


@pytest.mark.parametrize("name", VC_CASES)
def test_spectrogram_matches_reference(golden_dir, name):
    rec = _load(golden_dir, name)
    spec = vc_oracle.spectrogram(rec["wave"])
    assert spec.shape == rec["spec"].shape
    assert (spec - rec["spec"]).abs().max().item() <= 1e-4 * rec["spec"].abs().max().item()


@pytest.mark.parametrize("name", VC_CASES)
def test_voice_conversion_matches_reference(golden_dir, synth_sd, name):
    rec = _load(golden_dir, name)
    case = rec["case"]
    torch.set_num_threads(8)
    with torch.no_grad():
        o_hat, mask, (z, z_p, z_hat) = vc_oracle.voice_conversion(
            case_state_dict(rec, synth_sd), CONVERTER_MODEL_CONFIG, rec["spec"], rec["lengths"], rec["g_src"],
            rec["g_tgt"], case["tau"], rec["noise"], zero_g=case["zero_g"])
    assert torch.equal(mask, rec["y_mask"])
    # latent tolerances scale with the latents' magnitude (the stress case runs at |z| ~ 17 instead of ~ 4)
    s = max(1.0, rec["z"].abs().max().item() / 4.0)
    for got, key, tol in ((z, "z", 2e-5 * s), (z_p, "z_p", 5e-5 * s), (z_hat, "z_hat", 1e-4 * s), (o_hat, "o_hat", 2e-5)):
        err = (got - rec[key]).abs().max().item()
        assert err <= tol, (key, err)
    if case["lengths"]:
        for b, n in enumerate(case["lengths"]):
            assert z_hat[b, :, n:].abs().max().item() == 0.0 if n < z_hat.shape[2] else True


@pytest.mark.parametrize("name", VC_CASES + ["ref_enc_b2_t200"])
def test_reference_encoder_matches_reference(golden_dir, synth_sd, name):
    rec = _load(golden_dir, name)
    with torch.no_grad():       # (the stress variant leaves ref_enc.* untouched)
        se = vc_oracle.reference_encoder(synth_sd, rec["spec"].transpose(1, 2))
    assert se.shape == rec["ref_enc"].shape
    assert (se - rec["ref_enc"]).abs().max().item() <= 2e-5


def test_param_spec_matches_reference_schema(golden_dir):
    from openvoice_amd.params import converter_param_spec
    schema = torch.load(os.path.join(golden_dir, "converter_state_dict_schema.pt"), weights_only=False)
    spec = converter_param_spec(513, **CONVERTER_MODEL_CONFIG)
    assert set(spec) == set(schema)
    for key, shape in schema.items():
        assert tuple(spec[key]) == tuple(shape), key


def benchmark_length_inputs(rec):
    """Inputs of the compact T = 861 fixture, regenerated from its seeds exactly as oracle/make_golden.py drew them."""
    from oracle.make_golden import synth_wave
    seed, t = rec["seed"], rec["case"]["frames"]
    wave = synth_wave(1, 256 * t, seed)
    gen = torch.Generator().manual_seed(seed + 1)
    g_src = 0.3 * torch.randn((1, 256, 1), generator=gen)
    g_tgt = 0.3 * torch.randn((1, 256, 1), generator=gen)
    noise = torch.randn(1, 192, t, generator=gen)
    return wave, g_src, g_tgt, noise


def test_voice_conversion_matches_reference_at_benchmark_length(golden_dir, synth_sd):
    """T = 861 frames (10 s @ 22.05 kHz, the length BASELINE.json's metric is quoted on), output of the
    unmodified reference; the fixture stores o_hat in full and per-channel sums of spec / z / z_p / z_hat."""
    rec = _load(golden_dir, "vc_b1_t861_benchmark_length")
    wave, g_src, g_tgt, noise = benchmark_length_inputs(rec)
    torch.set_num_threads(8)
    with torch.no_grad():
        spec = vc_oracle.spectrogram(wave)
        o_hat, _, (z, z_p, z_hat) = vc_oracle.voice_conversion(
            synth_sd, CONVERTER_MODEL_CONFIG, spec, torch.tensor([spec.shape[2]]), g_src, g_tgt, rec["case"]["tau"],
            noise, zero_g=rec["case"]["zero_g"])
    assert spec.shape[2] == 861
    assert (spec.sum(2) - rec["spec_sum"]).abs().max().item() <= 1e-4 * rec["spec_sum"].abs().max().item()
    for got, key in ((z, "z_sum"), (z_p, "z_p_sum"), (z_hat, "z_hat_sum")):
        assert (got.sum(2) - rec[key]).abs().max().item() <= 5e-3, key       # sums of 861 values of |z| ~ 1
    assert (o_hat - rec["o_hat"]).abs().max().item() <= 2e-5
