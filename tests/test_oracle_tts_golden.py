"""Pins the TTS oracle (oracle/tts_oracle.py) to outputs of the unmodified reference
``SynthesizerTrn.infer`` and its parts (fixtures tests/golden/tts_*.pt, made by oracle/make_golden.py
in the build container).  Tolerances are fp32 round-off of different summation orders; the decoder
output is checked at 2e-5, the flow latents (|z| up to ~25 with the synthetic weights) at 2e-4."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import tts_oracle
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG

TTS_CASES = ["tts_b3_tx23_ragged", "tts_b1_tx40_slow", "tts_b2_tx5_tiny"]


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)


def test_tts_param_spec_matches_reference_schema(golden_dir):
    from openvoice_amd.params import tts_full_param_spec
    schema = torch.load(os.path.join(golden_dir, "tts_state_dict_schema.pt"), weights_only=False)
    spec = tts_full_param_spec(68, 10, 513, **CFG)
    assert set(spec) == set(schema)
    for key, shape in schema.items():
        assert tuple(spec[key]) == tuple(shape), key


@pytest.mark.parametrize("name", TTS_CASES)
def test_text_encoder_and_duration_predictors_match_reference(golden_dir, synth_tts_sd, name):
    rec = _load(golden_dir, name)
    sd, case = synth_tts_sd, rec["case"]
    with torch.no_grad():
        x, m, logs, mask = tts_oracle.text_encoder(sd, CFG, rec["tokens"], rec["lengths"])
        g = F.embedding(rec["sid"], sd["emb_g.weight"]).unsqueeze(-1)
        dp = tts_oracle.duration_predictor(sd, x, mask, g)
        sdp = tts_oracle.stochastic_duration_predictor_reverse(sd, x, mask, g, rec["noise_w"], case["noise_scale_w"])
    assert torch.equal(mask, rec["x_mask"])
    for got, key in ((x, "x"), (m, "m_tok"), (logs, "logs_tok"), (dp, "logw_dp"), (sdp, "logw_sdp")):
        err = (got - rec[key]).abs().max().item()
        assert err <= 2e-5, (key, err)


@pytest.mark.parametrize("name", TTS_CASES)
def test_infer_matches_reference(golden_dir, synth_tts_sd, name):
    rec = _load(golden_dir, name)
    case = rec["case"]
    torch.set_num_threads(8)
    with torch.no_grad():
        o, attn, y_mask, (z, z_p, m_p, logs_p), _ = tts_oracle.infer(
            synth_tts_sd, CFG, rec["tokens"], rec["lengths"], rec["sid"], rec["noise_w"], rec["noise_z"],
            noise_scale=case["noise_scale"], length_scale=case["length_scale"],
            noise_scale_w=case["noise_scale_w"], sdp_ratio=case["sdp_ratio"])
    assert torch.equal(attn, rec["attn"]) and torch.equal(y_mask, rec["y_mask"])
    for got, key, tol in ((m_p, "m_p", 2e-5), (logs_p, "logs_p", 2e-5), (z_p, "z_p", 2e-4), (z, "z", 2e-4),
                          (o, "o", 2e-5)):
        err = (got - rec[key]).abs().max().item()
        assert err <= tol, (key, err)


def test_spline_inverse_round_trips_the_forward_spline():
    """Size-independent property: the inverse restated here undoes the textbook forward
    rational-quadratic map (monotone in each bin), including the linear tails."""
    gen = torch.Generator().manual_seed(0)
    n, nb, tb = 4096, 10, 5.0
    uw, uh, ud = torch.randn(n, nb, generator=gen), torch.randn(n, nb, generator=gen), torch.randn(n, nb - 1, generator=gen)
    y = (torch.rand(n, generator=gen) * 14 - 7)
    x = tts_oracle.rq_spline_inverse(y, uw, uh, ud)
    # forward map built independently from knots
    import math
    w = 1e-3 + (1 - 1e-3 * nb) * torch.softmax(uw, -1)
    h = 1e-3 + (1 - 1e-3 * nb) * torch.softmax(uh, -1)
    cw = F.pad(torch.cumsum(w, -1), (1, 0)) * 2 * tb - tb
    ch = F.pad(torch.cumsum(h, -1), (1, 0)) * 2 * tb - tb
    cw[:, 0], cw[:, -1], ch[:, 0], ch[:, -1] = -tb, tb, -tb, tb
    d = 1e-3 + F.softplus(F.pad(ud, (1, 1), value=math.log(math.exp(1 - 1e-3) - 1)))
    inside = (x > -tb) & (x < tb)
    k = ((x[:, None] >= cw).sum(-1) - 1).clamp(0, nb - 1)[:, None]
    bw, bh = (cw[:, 1:] - cw[:, :-1]).gather(1, k)[:, 0], (ch[:, 1:] - ch[:, :-1]).gather(1, k)[:, 0]
    th = (x - cw.gather(1, k)[:, 0]) / bw
    s, d0, d1 = bh / bw, d.gather(1, k)[:, 0], d.gather(1, k + 1)[:, 0]
    fwd = ch.gather(1, k)[:, 0] + bh * (s * th * th + d0 * th * (1 - th)) / (s + (d0 + d1 - 2 * s) * th * (1 - th))
    fwd = torch.where(inside, fwd, x)
    assert (fwd - y).abs().max().item() <= 2e-4
    assert torch.equal(x[~((y >= -tb) & (y <= tb))], y[~((y >= -tb) & (y <= tb))])
