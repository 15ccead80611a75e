"""Length-aware work lists (``ov_conv1d_params.col_limit``, ``ov_conv_post_tanh_limited_f32``, ``ov_frame_limits_i32``;
VERDICT r02 item 7): a padded batch -- ragged ``convert_batch``, batched TTS, reference openvoice/models.py:477-489 pads
every utterance to the longest -- costs the generator only ``length + 16`` frames per utterance.  What is computed is
bit-identical to the full launch; what is skipped is not written (conv) / written as zero (conv_post)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import PackedConv, launch_conv, launch_pair  # noqa: E402
from openvoice_amd.models import SynthesizerTrn  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_frame_limits():
    lengths = torch.tensor([0, 5, 100, 845, 861, 2000, -3], dtype=torch.int64, device=DEV)
    limits = torch.full((7,), -1, dtype=torch.int32, device=DEV)
    _lib.call("ov_frame_limits_i32", lengths, limits, 7, 861, 16)
    assert limits.tolist() == [16, 21, 116, 861, 861, 861, 16]


@pytest.mark.parametrize("c,k,d,L,tpw,tile_cols", [(256, 3, 1, 1900, 0, 128), (256, 11, 5, 1400, -1, 128),
                                                   (128, 7, 3, 3000, -1, 128), (64, 3, 1, 5000, 0, 256),
                                                   (32, 11, 1, 9000, -1, 512), (128, 3, 1, 700, 2, 128)])
def test_limited_conv_computes_the_same_bits_and_skips_the_rest(c, k, d, L, tpw, tile_cols):
    """Utterances with limits 0, 1, mid-tile, a tile boundary, > L and a scale of 2: every column of a time tile that
    starts before the limit equals the full launch bit for bit; tiles that start at or beyond it are not written
    (the NaN poison stays).  Persistent, one-tile-per-workgroup and forced tiles-per-workgroup launches; one and two
    M-blocks per time tile; with a residual operand."""
    B = 6
    if c > 64 and B * -(-c // 128) * -(-L // 128) < torch.cuda.get_device_properties(0).multi_processor_count:
        tile_cols = 256      # small-launch rule (ov_api.hip): fewer 128 x 128 tiles than CUs -> the 32 x 256 tile
    x, res = _rand(B, c, L, seed=1).to(DEV), _rand(B, c, L, seed=2).to(DEV)
    layer = PackedConv(_rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1), DEV, K=k, dil=d)
    full = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, x, 0, c * L, full, 0, c * L, B, L, in_slope=0.1, res=res, res_bs=c * L, tiles_per_wg=tpw)
    scale = 2
    cols = [0, 1, L // 3 + 7, 2 * tile_cols, L + 10, L // 2]            # valid output columns per utterance
    limits = torch.tensor([(n + scale - 1) // scale for n in cols], dtype=torch.int32, device=DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=res, res_bs=c * L, tiles_per_wg=tpw,
                col_limit=limits, col_limit_scale=scale)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    for b, lim in enumerate((limits.cpu() * scale).clamp(max=L).tolist()):
        computed = min(L, -(-lim // tile_cols) * tile_cols)           # whole tiles that start before the limit
        assert torch.equal(out[b, :, :computed], full[b, :, :computed]), f"utterance {b}: computed columns differ"
        assert torch.isnan(out[b, :, computed:]).all(), f"utterance {b}: columns beyond tile {computed} were written"


@pytest.mark.parametrize("c,k,d,L,nwg,nt", [(32, 3, 1, 9000, 0, 256), (32, 7, 5, 5000, 7, 256), (64, 3, 3, 3000, 0, 128),
                                              (32, 11, 3, 2600, 3, 256), (64, 3, 1, 1000, 100000, 128)])
def test_limited_fused_pair_computes_the_same_bits_and_skips_the_rest(c, k, d, L, nwg, nt):
    """The fused ResBlock pair (sliding window along time) with per-utterance limits: utterance b is walked for
    ceil((limit + P2) / NT) steps -- none at limit 0 -- so the columns [0, steps * NT - P2) equal the full launch bit for
    bit and nothing beyond is written; runs that start mid-utterance and span utterances of different step counts
    (forced workgroup counts), with the MRF addend."""
    B, p2 = 6, (k - 1) // 2
    gen_w = lambda seed, scale: _rand(c, c, k, seed=seed, scale=scale * (c * k) ** -0.5)
    c1 = PackedConv(gen_w(3, 1.0), _rand(c, seed=4, scale=0.1), DEV, K=k, dil=d)
    c2 = PackedConv(gen_w(5, 0.5), _rand(c, seed=6, scale=0.1), DEV, K=k, dil=1)
    x, add = _rand(B, c, L, seed=1).to(DEV), _rand(B, c, L, seed=2).to(DEV)
    full = torch.full((B, c, L), float("nan"), device=DEV)
    launch_pair(c1, c2, x, c * L, full, c * L, B, L, add=add, add_bs=c * L, scale=0.5, nwg=nwg)
    cols = [0, 1, L // 3 + 7, 2 * nt, L + 10, L // 2]
    limits = torch.tensor(cols, dtype=torch.int32, device=DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_pair(c1, c2, x, c * L, out, c * L, B, L, add=add, add_bs=c * L, scale=0.5, nwg=nwg, col_limit=limits,
                col_limit_scale=1)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    for b, lim in enumerate(min(n, L) for n in cols):
        steps = -(-(lim + p2) // nt) if lim > 0 else 0
        computed = max(0, min(L, steps * nt - p2))
        assert computed >= lim
        assert torch.equal(out[b, :, :computed], full[b, :, :computed]), f"utterance {b}: computed columns differ"
        assert torch.isnan(out[b, :, computed:]).all(), f"utterance {b}: columns beyond {computed} were written"


def test_limit_argument_checks():
    c, L = 32, 600
    layer = PackedConv(_rand(c, c, 3, seed=3), _rand(c, seed=4), DEV, K=3)
    x, out = torch.zeros(2, c, L, device=DEV), torch.zeros(2, c, L, device=DEV)
    lim = torch.tensor([10, 20], dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.OvError, match="BADARG"):
        launch_conv(layer, x, 0, c * L, out, 0, c * L, 2, L, col_limit=lim, col_limit_scale=0)
    if _lib.use_torch_binding():
        with pytest.raises(_lib.OvError, match="int32"):
            launch_conv(layer, x, 0, c * L, out, 0, c * L, 2, L, col_limit=lim.long(), col_limit_scale=1)


def test_conv_post_limited():
    B, C, L = 3, 32, 4096
    x = _rand(B, C, L, seed=5).to(DEV)
    w = _rand(C, 7, seed=6, scale=0.1).to(DEV)
    full = torch.empty(B, 1, L, device=DEV)
    _lib.call("ov_conv_post_tanh_f32", x, w, full, B, C, L, 7, 0.01)
    lim = torch.tensor([0, 5, 100], dtype=torch.int32, device=DEV)
    out = torch.full((B, 1, L), float("nan"), device=DEV)
    _lib.call("ov_conv_post_tanh_limited_f32", x, w, out, B, C, L, 7, 0.01, lim, 256)
    torch.cuda.synchronize()
    for b, n in enumerate([0, 1280, 4096]):
        assert torch.equal(out[b, :, :n], full[b, :, :n]) and (out[b, :, n:] == 0).all()


def _model(sd, zero_g=True):
    m = SynthesizerTrn(0, 513, n_speakers=0, zero_g=zero_g, **CONVERTER_MODEL_CONFIG)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def test_voice_conversion_skip_padding_keeps_every_valid_sample(synth_sd):
    """Ragged batch 300 / 171 / 40 / 1 frames: with ``skip_padding`` the first ``length`` frames of every utterance
    (``256 * length`` samples) are ``torch.equal`` to the full computation -- the generator's receptive field (13.3
    frames) fits the 16-frame margin -- the latents are untouched, and the tail beyond the computed tiles is zero."""
    B, T = 4, 300
    gen = torch.Generator().manual_seed(31)
    spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(DEV)
    lengths = torch.tensor([300, 171, 40, 1], device=DEV)
    g1, g2 = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV), (0.3 * torch.randn(B, 256, 1, generator=gen)).to(DEV)
    noise = torch.randn(B, 192, T, generator=gen).to(DEV)
    model = _model(synth_sd)
    o_full, m_full, lat_full = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise)
    o_skip, m_skip, lat_skip = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise, skip_padding=True)
    torch.cuda.synchronize()
    assert torch.equal(m_full, m_skip) and all(torch.equal(a, b) for a, b in zip(lat_full, lat_skip))
    assert torch.isfinite(o_skip).all()
    for b, n in enumerate(lengths.tolist()):
        assert torch.equal(o_skip[b, :, :256 * n], o_full[b, :, :256 * n]), f"utterance {b}: valid samples changed"
        # beyond the utterance: silence (conv_post keeps ``length`` frames; the generator computed length + margin)
        if n < T:
            assert (o_skip[b, :, 256 * n:] == 0).all() and o_full[b, :, 256 * n:].abs().max() > 0
    # graph replay of the same shape takes the same path
    o_graph = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise, graph=True, skip_padding=True)[0]
    assert torch.equal(o_graph, o_skip)


def test_tts_infer_skip_padding_keeps_every_valid_sample(synth_tts_sd):
    """Batched TTS (BASELINE.json configs[3] shape, scaled down): ``infer(..., skip_padding=True)`` returns the same
    samples within each utterance's own length as the padded computation."""
    model = SynthesizerTrn(68, 513, n_speakers=10, **CONVERTER_MODEL_CONFIG)
    model.load_state_dict(synth_tts_sd, strict=True)
    model = model.to(DEV).eval()
    gen = torch.Generator().manual_seed(17)
    lengths = torch.tensor([40, 25, 9, 33])
    tokens = torch.randint(0, 68, (4, 40), generator=gen)
    sid = torch.tensor([0, 3, 5, 9])
    noise_w = torch.randn(4, 2, 40, generator=gen)
    noise_z = torch.randn(4, 192, 1200, generator=gen)
    kw = dict(sid=sid.to(DEV), noise_scale=0.667, noise_scale_w=0.6, length_scale=1.0, noise_w=noise_w.to(DEV),
              noise_z=noise_z.to(DEV))
    o_full, attn, y_mask, _ = model.infer(tokens.to(DEV), lengths.to(DEV), **kw)
    o_skip, attn2, y_mask2, _ = model.infer(tokens.to(DEV), lengths.to(DEV), skip_padding=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(attn, attn2) and torch.equal(y_mask, y_mask2) and o_full.shape == o_skip.shape
    frames = y_mask[:, 0].sum(1).long().tolist()
    assert len(set(frames)) > 1, "the batch must be ragged in frames for this test to mean anything"
    for b, n in enumerate(frames):
        assert torch.equal(o_skip[b, :, :256 * n], o_full[b, :, :256 * n]), f"utterance {b}"
        assert (o_skip[b, :, 256 * n:] == 0).all()


def test_large_batches_at_and_beyond_the_prefix_table(synth_sd):
    """B = 256 is the most utterances the kernels' per-workgroup prefix table holds; B = 257 must fall back to whole
    tensors (``skip_padding`` is an optimisation, never a correctness requirement).  Short utterances (T = 40) keep the
    288 GB card far from full; items are compared with their own batch-1 conversion (utterances are independent)."""
    T = 40
    gen = torch.Generator().manual_seed(256)
    model = _model(synth_sd)
    g = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV)
    for B in (256, 257):
        spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(DEV)
        lengths = torch.randint(1, T + 1, (B,), generator=gen).to(DEV)
        lengths[0] = T
        noise = torch.randn(B, 192, T, generator=gen).to(DEV)
        o = model.voice_conversion(spec, lengths, g, g, tau=0.3, noise=noise, skip_padding=True)[0]
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        for b in (0, 1, B // 2, B - 1):
            n = int(lengths[b])
            one = model.voice_conversion(spec[b:b + 1, :, :], lengths[b:b + 1], g, g, tau=0.3, noise=noise[b:b + 1])[0]
            # (not bit for bit: the launch policy picks the conv algorithm by launch size, engine.WINO_MIN_ITEMS -- a
            # batch-1 conversion of 40 frames runs the direct kernels where the batch of 256 runs the Winograd-domain ones)
            assert (o[b, :, :256 * n] - one[0, :, :256 * n]).abs().max().item() <= 1e-4, (B, b)
            if B <= 256:
                assert (o[b, :, 256 * n:] == 0).all()


def test_one_minute_utterances(synth_sd):
    """T = 5 168 frames (60 s): 1.3 M samples per utterance, every 32-bit offset computation of the kernels at 6x the
    benchmark length; against the oracle on the first and the last second and through the flow's round trip."""
    from oracle import vc_oracle
    B, T = 2, 5168
    gen = torch.Generator().manual_seed(60)
    spec = torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]
    lengths = torch.tensor([T, T - 999])
    g1, g2 = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
    noise = torch.randn(B, 192, T, generator=gen)
    model = _model(synth_sd)
    o, _, (z, z_p, z_hat) = model.voice_conversion(spec.to(DEV), lengths.to(DEV), g1.to(DEV), g2.to(DEV), tau=0.3,
                                                   noise=noise.to(DEV))
    o_same, _, (_, _, z_rt) = model.voice_conversion(spec.to(DEV), lengths.to(DEV), g1.to(DEV), g1.to(DEV), tau=0.3,
                                                     noise=noise.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(o).all() and o.shape == (B, 1, 256 * T)
    assert (z_rt - z).abs().max().item() <= 2e-4          # same speaker both ways: the flow inverts itself
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        o_ref = vc_oracle.voice_conversion(synth_sd, CONVERTER_MODEL_CONFIG, spec, lengths, g1, g2, 0.3, noise,
                                           zero_g=True)[0]
    err = (o.cpu() - o_ref).abs().max().item()
    print("60 s utterances vs oracle:", err)
    assert err <= 1e-3
