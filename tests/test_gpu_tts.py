"""Parity of the V1 TTS path on the MI355X: every token-rate kernel of csrc/tts.hip against the same op in
plain fp32 PyTorch (through the C ABI), then ``SynthesizerTrn.infer`` end to end against the committed
reference outputs (tests/golden/tts_*.pt, produced by the unmodified reference) and the oracle.

Tolerances: fp32 VALU/MFMA arithmetic with a different summation order, so 2e-5 relative on O(1) tensors;
the waveform bound is BASELINE.json's 1e-3 max-abs.  Durations (ceil of exp(logw)) must match exactly."""
import ctypes
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import _ptr  # noqa: E402
from openvoice_amd.models import SynthesizerTrn  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG  # noqa: E402

DEV = "cuda:0"
TTS_CASES = ["tts_b3_tx23_ragged", "tts_b1_tx40_slow", "tts_b2_tx5_tiny"]


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _close(got, ref, rel=2e-5, what=""):
    err = (got.cpu() - ref).abs().max().item()
    bound = rel * max(1.0, ref.abs().max().item())
    assert err <= bound, f"{what}: max-abs err {err:.3e} > {bound:.3e}"


def _padded(t, ld):
    out = torch.full(t.shape[:-1] + (ld,), float("nan"))
    out[..., :t.shape[-1]] = t
    return out


def _mask(lengths, T):
    return (torch.arange(T)[None] < torch.tensor(lengths)[:, None]).float()


@pytest.mark.parametrize("T", [1, 7, 200])
def test_embed_and_layernorm(T):
    lib = _lib.load()
    B, H, V, ld = 2, 192, 68, (T + 3) // 4 * 4
    gen = torch.Generator().manual_seed(T)
    tok = torch.randint(0, V, (B, T), generator=gen)
    lens = [T, max(1, T - 3)]
    emb = _rand(V, H, seed=1)
    ref = (F.embedding(tok, emb) * math.sqrt(H)).transpose(1, 2) * _mask(lens, T)[:, None]
    out = torch.full((B, H, ld), float("nan"), device=DEV)
    tokd, embd, lend = tok.to(DEV), emb.to(DEV), torch.tensor(lens, device=DEV)
    assert lib.ov_embed_f32(ctypes.c_void_p(tokd.data_ptr()), _ptr(embd), ctypes.c_void_p(lend.data_ptr()), _ptr(out),
                            B, T, H, V, ld, math.sqrt(H), _st()) == 0
    _close(out[:, :, :T], ref, what="embed")
    # LayerNorm: residual in, relu before, gelu after, residual after, mask
    x, res, res2 = _rand(B, H, T, seed=2), _rand(B, H, T, seed=3), _rand(B, H, T, seed=4)
    gamma, beta = 1 + 0.1 * _rand(H, seed=5), 0.1 * _rand(H, seed=6)
    mask = _mask(lens, T)
    ln = lambda v: F.layer_norm(v.transpose(1, 2), (H,), gamma, beta, 1e-5).transpose(1, 2)
    cases = [(dict(res=res, mask=mask), 0, ln(x + res) * mask[:, None]),
             (dict(mask=mask), _lib.LN_PRE_RELU, ln(torch.relu(x)) * mask[:, None]),
             (dict(), _lib.LN_POST_GELU, F.gelu(ln(x))),
             (dict(res2=res2, mask=mask), _lib.LN_POST_GELU, (res2 + F.gelu(ln(x))) * mask[:, None])]
    for kw, flags, want in cases:
        dv = lambda t: _padded(t, ld).to(DEV) if t is not None else None
        xd, rd, r2d, md = dv(x), dv(kw.get("res")), dv(kw.get("res2")), dv(kw.get("mask"))
        outd = torch.full((B, H, ld), float("nan"), device=DEV)
        assert lib.ov_layernorm_ch_f32(_ptr(xd), _ptr(rd) if rd is not None else None, _ptr(gamma.to(DEV)),
                                       _ptr(beta.to(DEV)), _ptr(r2d) if r2d is not None else None,
                                       _ptr(md) if md is not None else None, _ptr(outd), B, H, T, ld, 1e-5, flags,
                                       _st()) == 0
        _close(outd[:, :, :T], want, what=f"layernorm flags={flags} {sorted(kw)}")
        assert torch.isnan(outd[:, :, T:]).all()


@pytest.mark.parametrize("T,lens", [(1, [1]), (3, [3, 2]), (23, [23, 15, 7]), (130, [130, 77])])
def test_relative_attention(T, lens):
    """Scores + relative keys on the band, -1e4 masking, softmax, values + relative values
    (reference: openvoice/attentions.py:264-329), against the oracle's direct restatement."""
    from oracle import tts_oracle
    lib = _lib.load()
    B, H, heads, w, ld = len(lens), 192, 2, 4, (T + 3) // 4 * 4
    dk = H // heads
    x = _rand(B, H, T, seed=1)
    sd = {f"a.conv_{c}.weight": _rand(H, H, 1, seed=2 + i, scale=H ** -0.5) for i, c in enumerate("qkvo")}
    sd.update({f"a.conv_{c}.bias": _rand(H, seed=6 + i, scale=0.1) for i, c in enumerate("qkvo")})
    sd["a.emb_rel_k"], sd["a.emb_rel_v"] = _rand(1, 2 * w + 1, dk, seed=10, scale=dk ** -0.5), _rand(1, 2 * w + 1, dk, seed=11, scale=dk ** -0.5)
    # reference output without the conv_o projection: use identity o
    sd["a.conv_o.weight"], sd["a.conv_o.bias"] = torch.eye(H)[:, :, None], torch.zeros(H)
    mask = _mask(lens, T)
    want = tts_oracle.relative_attention(sd, "a", x, mask[:, None], heads)
    qkv = torch.cat([F.conv1d(x, sd[f"a.conv_{c}.weight"], sd[f"a.conv_{c}.bias"]) for c in "qkv"], 1)
    qkvd = _padded(qkv, ld).to(DEV)
    outd = torch.full((B, H, ld), float("nan"), device=DEV)
    assert lib.ov_rel_attention_f32(_ptr(qkvd), _ptr(qkvd, H * ld), _ptr(qkvd, 2 * H * ld),
                                    _ptr(sd["a.emb_rel_k"][0].contiguous().to(DEV)),
                                    _ptr(sd["a.emb_rel_v"][0].contiguous().to(DEV)), _ptr(_padded(mask, ld).to(DEV)),
                                    _ptr(outd), 3 * H * ld, H * ld, B, heads, dk, T, ld, w, _st()) == 0
    for b, n in enumerate(lens):     # rows of padded queries are unspecified (masked downstream)
        _close(outd[b, :, :n], want[b, :, :n], what=f"attention T={T} b={b}")


def test_dwconv_expand1_add_bias():
    lib = _lib.load()
    B, C, T, ld = 2, 192, 77, 80
    x, g = _rand(B, C, T, seed=1), _rand(B, C, T, seed=2)
    mask = _mask([T, 40], T)
    for dil in (1, 3, 9):
        w, b = _rand(C, 1, 3, seed=3), _rand(C, seed=4)
        want = F.conv1d(x * mask[:, None], w, b, padding=dil, dilation=dil, groups=C)
        outd = torch.full((B, C, ld), float("nan"), device=DEV)
        assert lib.ov_dwconv1d_f32(_ptr(_padded(x, ld).to(DEV)), _ptr(w[:, 0].contiguous().to(DEV)), _ptr(b.to(DEV)),
                                   _ptr(_padded(mask, ld).to(DEV)), _ptr(outd), B, C, T, ld, 3, dil, _st()) == 0
        _close(outd[:, :, :T], want, what=f"dwconv dil={dil}")
    z = _rand(B, 2, T, seed=5)
    w1, b1 = _rand(C, seed=6), _rand(C, seed=7)
    zd = _padded(z, ld).to(DEV)
    outd = torch.full((B, C, ld), float("nan"), device=DEV)
    assert lib.ov_expand1_f32(_ptr(zd, ld), 2 * ld, _ptr(w1.to(DEV)), _ptr(b1.to(DEV)), _ptr(_padded(g, ld).to(DEV)),
                              _ptr(outd), B, C, T, ld, _st()) == 0
    _close(outd[:, :, :T], w1[None, :, None] * z[:, 1:2] + b1[None, :, None] + g, what="expand1")
    bias_b = _rand(B, C, seed=8)
    assert lib.ov_add_bias_mask_f32(_ptr(_padded(x, ld).to(DEV)), _ptr(bias_b.to(DEV)), _ptr(_padded(mask, ld).to(DEV)),
                                    _ptr(outd), B, C, T, ld, _st()) == 0
    _close(outd[:, :, :T], (x + bias_b[:, :, None]) * mask[:, None], what="add_bias_mask")


@pytest.mark.parametrize("c0,c1", [(0, 1), (1, 0)])
def test_rq_spline_inverse(c0, c1):
    """Inverse rational-quadratic spline incl. both linear tails and values exactly on a knot
    (reference: openvoice/transforms.py:50-188), against the oracle's per-element restatement."""
    from oracle import tts_oracle
    lib = _lib.load()
    B, T, ld, F_ = 2, 1000, 1000, 192
    z = _rand(B, 2, T, seed=1, scale=3.0)
    z[0, c1, :4] = torch.tensor([-5.0, 5.0, 0.0, 7.5])
    h = _rand(B, 32, T, seed=2, scale=8.0)
    mask = _mask([T, 600], T)
    hm = (h * mask[:, None]).transpose(1, 2)
    want1 = tts_oracle.rq_spline_inverse(z[:, c1], hm[..., :10] / math.sqrt(F_), hm[..., 10:20] / math.sqrt(F_),
                                         hm[..., 20:29])
    zd = z.to(DEV).contiguous()
    assert lib.ov_rq_spline_inverse_f32(_ptr(zd), 2 * ld, c0, c1, _ptr((h * mask[:, None]).to(DEV)), 32 * ld,
                                        _ptr(mask.to(DEV)), B, T, ld, 10, F_, 5.0, _st()) == 0
    _close(zd[:, c1], want1 * mask, rel=1e-4, what="spline x1")
    _close(zd[:, c0], z[:, c0] * mask, what="spline x0 (masked only)")


def test_duration_and_expand_prior():
    lib = _lib.load()
    B, Tx, Lx, C = 3, 11, 12, 192
    lens = [11, 6, 1]
    mask = _mask(lens, Tx)
    z_sdp, dp = _rand(B, 2, Tx, seed=1), _rand(B, 32, Tx, seed=2) * mask[:, None]
    m, lg, ratio, ls = 0.1, -0.2, 0.2, 1.1
    logw = ((z_sdp[:, 0] - m) * math.exp(-lg) * mask) * ratio + dp[:, 0] * (1 - ratio)
    w = torch.ceil(torch.exp(logw) * mask * ls)
    cum_ref = torch.cumsum(w, 1).to(torch.int32)
    ylen_ref = w.sum(1).clamp_min(1).long()
    logwd = torch.full((B, Lx), float("nan"), device=DEV)
    cumd = torch.zeros(B, Lx, dtype=torch.int32, device=DEV)
    ylend = torch.zeros(B, dtype=torch.int64, device=DEV)
    assert lib.ov_duration_f32(_ptr(_padded(z_sdp, Lx).to(DEV)), 2 * Lx, m, lg, _ptr(_padded(dp, Lx).to(DEV)), 32 * Lx,
                               _ptr(_padded(mask, Lx).to(DEV)), _ptr(logwd), ctypes.c_void_p(cumd.data_ptr()),
                               ctypes.c_void_p(ylend.data_ptr()), B, Tx, Lx, ratio, ls, _st()) == 0
    _close(logwd[:, :Tx], logw, what="logw")
    assert torch.equal(cumd[:, :Tx].cpu(), cum_ref) and torch.equal(ylend.cpu(), ylen_ref)
    # expansion
    Ty = int(ylen_ref.max())
    Ly = (Ty + 3) // 4 * 4
    stats = _rand(B, 2 * C, Tx, seed=3) * mask[:, None]
    noise = _rand(B, C, Ty, seed=4)
    ymask = _mask(ylen_ref.tolist(), Ty)
    path = torch.zeros(B, Ty, Tx)
    for b in range(B):
        lo = 0
        for j in range(lens[b]):
            hi = int(cum_ref[b, j])
            path[b, lo:min(hi, Ty), j] = 1
            lo = hi
    path = path * ymask[:, :, None]
    m_p = (path @ stats[:, :C].transpose(1, 2)).transpose(1, 2)
    logs_p = (path @ stats[:, C:].transpose(1, 2)).transpose(1, 2)
    z_p = m_p + noise * torch.exp(logs_p) * 0.667
    zpd, mpd, lpd = (torch.full((B, C, Ly), float("nan"), device=DEV) for _ in range(3))
    attnd = torch.full((B, Ty, Tx), float("nan"), device=DEV)
    lend = torch.tensor(lens, device=DEV)
    statsd = _padded(stats, Lx).to(DEV)
    assert lib.ov_expand_prior_f32(_ptr(statsd), _ptr(statsd, C * Lx), 2 * C * Lx, Lx, ctypes.c_void_p(cumd.data_ptr()),
                                   ctypes.c_void_p(lend.data_ptr()), ctypes.c_void_p(ylend.data_ptr()),
                                   _ptr(_padded(noise, Ly).to(DEV)), C * Ly, Ly, _ptr(zpd), _ptr(mpd), _ptr(lpd),
                                   _ptr(attnd), B, C, Tx, Ty, Ly, 0.667, _st()) == 0
    assert torch.equal(attnd.cpu(), path)
    _close(mpd[:, :, :Ty], m_p, what="m_p")
    _close(lpd[:, :, :Ty], logs_p, what="logs_p")
    _close(zpd[:, :, :Ty], z_p, what="z_p")


def _tts_model(sd):
    m = SynthesizerTrn(68, 513, n_speakers=10, **CFG)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name", TTS_CASES)
def test_infer_matches_reference_golden(golden_dir, synth_tts_sd, name):
    rec = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    case = rec["case"]
    model = _tts_model(synth_tts_sd)
    o, attn, y_mask, (z, z_p, m_p, logs_p) = model.infer(
        rec["tokens"].to(DEV), rec["lengths"].to(DEV), sid=rec["sid"].to(DEV), noise_scale=case["noise_scale"],
        length_scale=case["length_scale"], noise_scale_w=case["noise_scale_w"], sdp_ratio=case["sdp_ratio"],
        noise_w=rec["noise_w"], noise_z=rec["noise_z"])
    torch.cuda.synchronize()
    logw_ref = rec["logw_sdp"] * case["sdp_ratio"] + rec["logw_dp"] * (1 - case["sdp_ratio"])
    errs = dict(logw=(model.engine().last_logw.cpu() - logw_ref[:, 0]).abs().max().item())
    assert torch.equal(attn.cpu(), rec["attn"]), "durations / alignment differ from the reference"
    assert torch.equal(y_mask.cpu(), rec["y_mask"])
    for got, key in ((m_p, "m_p"), (logs_p, "logs_p"), (z_p, "z_p"), (z, "z"), (o, "o")):
        errs[key] = (got.cpu() - rec[key]).abs().max().item()
    print(name, errs)
    assert errs["logw"] <= 1e-4 and errs["m_p"] <= 5e-5 and errs["logs_p"] <= 5e-5, errs
    assert errs["z_p"] <= 5e-4 and errs["z"] <= 5e-4, errs           # |z| up to ~25 with the synthetic weights
    assert errs["o"] <= 1e-3, errs
    assert o.shape == rec["o"].shape


def test_infer_matches_oracle_at_config4_shape(synth_tts_sd):
    """BASELINE.json configs[3]-like shape: batch 16, ~100 tokens per utterance, ragged; vs the CPU oracle."""
    from oracle import tts_oracle
    from openvoice_amd.hostinfo import usable_cpus
    gen = torch.Generator().manual_seed(9)
    B, Tx = 4, 101
    tokens = torch.randint(0, 68, (B, Tx), generator=gen)
    lengths = torch.tensor([101, 64, 33, 100])
    sid = torch.tensor([0, 1, 5, 9])
    noise_w = torch.randn(B, 2, Tx, generator=gen)
    noise_z = torch.randn(B, 192, 8 * Tx, generator=gen)
    torch.set_num_threads(usable_cpus(32))
    with torch.no_grad():
        o_r, attn_r, ym_r, (z_r, zp_r, mp_r, lp_r), _ = tts_oracle.infer(
            synth_tts_sd, CFG, tokens, lengths, sid, noise_w, noise_z, 0.667, 1.0, 0.6, 0.2)
    model = _tts_model(synth_tts_sd)
    o, attn, y_mask, (z, z_p, m_p, logs_p) = model.infer(tokens, lengths, sid=sid, noise_scale=0.667, length_scale=1.0,
                                                         noise_scale_w=0.6, sdp_ratio=0.2, noise_w=noise_w,
                                                         noise_z=noise_z)
    torch.cuda.synchronize()
    assert torch.equal(attn.cpu(), attn_r) and torch.equal(y_mask.cpu(), ym_r)
    errs = dict(z=(z.cpu() - z_r).abs().max().item(), z_p=(z_p.cpu() - zp_r).abs().max().item(),
                o=(o.cpu() - o_r).abs().max().item())
    print("config-4 shape", tuple(o.shape), errs)
    assert errs["z"] <= 5e-4 and errs["z_p"] <= 5e-4 and errs["o"] <= 1e-3, errs
    # max_len truncates only the decoder input (models.py:489)
    o2 = model.infer(tokens, lengths, sid=sid, noise_scale=0.667, noise_scale_w=0.6, noise_w=noise_w, noise_z=noise_z,
                     max_len=50)[0]
    assert o2.shape[2] == 50 * 256


def test_voice_conversion_on_a_tts_checkpoint(synth_tts_sd):
    """A TTS checkpoint also carries enc_q / flow / dec; voice_conversion must keep working on it."""
    from oracle import vc_oracle
    gen = torch.Generator().manual_seed(3)
    spec = torch.rand(1, 513, 20, generator=gen)
    g1, g2 = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
    noise = torch.randn(1, 192, 20, generator=gen)
    model = _tts_model(synth_tts_sd)
    o = model.voice_conversion(spec, torch.tensor([20]), g1, g2, tau=0.3, noise=noise)[0]
    with torch.no_grad():
        o_r = vc_oracle.voice_conversion(synth_tts_sd, CFG, spec, torch.tensor([20]), g1, g2, 0.3, noise)[0]
    assert (o.cpu() - o_r).abs().max().item() <= 1e-3


def test_base_speaker_tts_api_from_checkpoint_files(tmp_path, synth_tts_sd):
    """BaseSpeakerTTS(config.json, device).load_ckpt(checkpoint.pth) + tts_from_ids: the per-sentence loop of the
    reference (openvoice/api.py:78-94) and the one-batch form agree away from the unmasked-decoder tails."""
    import json
    import numpy as np
    from openvoice_amd import api
    from openvoice_amd.utils import CONVERTER_DATA_CONFIG
    cfg = {"data": dict(CONVERTER_DATA_CONFIG, n_speakers=10, text_cleaners=["cjke_cleaners2"], add_blank=True),
           "model": dict(CFG), "symbols": [f"s{i}" for i in range(68)], "speakers": {"default": 1, "whispering": 2}}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    torch.save({"model": synth_tts_sd}, tmp_path / "checkpoint.pth")
    tts = api.BaseSpeakerTTS(str(tmp_path / "config.json"), device=DEV)
    tts.load_ckpt(str(tmp_path / "checkpoint.pth"))
    gen = torch.Generator().manual_seed(1)
    ids = [api.intersperse(torch.randint(1, 68, (n,), generator=gen).tolist(), 0) for n in (12, 7, 20)]
    torch.manual_seed(0)
    single = tts.tts_from_ids(ids, tts.hps.speakers["default"], speed=1.0)
    torch.manual_seed(0)
    batch = tts.tts_from_ids(ids, tts.hps.speakers["default"], speed=1.0, batched=True)
    assert len(single) == len(batch) == 3 and all(a.dtype == np.float32 and a.ndim == 1 for a in single)
    assert all(len(a) % 256 == 0 and len(a) > 0 for a in single)
    # api.tts with a registered front end: text -> ids hook, sentence split, 50 ms gaps, WAV on disk
    api.BaseSpeakerTTS.text_to_sequence = staticmethod(lambda text, symbols, cleaners: [1 + (ord(c) % 67) for c in text])
    try:
        tts.tts("Hello world, this is a test. Another short one!", str(tmp_path / "out.wav"), speaker="default",
                language="English", speed=1.1)
    finally:
        api.BaseSpeakerTTS.text_to_sequence = None
    from openvoice_amd import audio_io
    wav, sr = audio_io.load(str(tmp_path / "out.wav"), 22050)
    assert sr == 22050 and len(wav) > 22050 // 4 and np.isfinite(wav).all()
