"""Load-time algebra of the engine, checked on CPU with plain PyTorch (no kernels involved): the rewrites that the
HIP path relies on must be exact re-expressions of the reference ops."""
import torch
import torch.nn.functional as F

from openvoice_amd.engine import conv_transpose_as_conv, convt_row_order, gate_row_order, padded_frames
from openvoice_amd.params import effective_weight


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_conv_transpose_equals_three_tap_phase_conv():
    """ConvTranspose1d(k = 2s, stride s, pad s/2) (reference: openvoice/models.py:244-256) == a 3-tap stride-1 conv
    whose row co*s + p is output phase p (engine.conv_transpose_as_conv), for every released (s, k)."""
    for s, cin, cout in ((8, 12, 6), (2, 10, 4)):
        x, w, b = _rand(2, cin, 23, seed=1), _rand(cin, cout, 2 * s, seed=2), _rand(cout, seed=3)
        ref = F.conv_transpose1d(x, w, b, stride=s, padding=s // 2)
        y = F.conv1d(x, conv_transpose_as_conv(w, s), b.repeat_interleave(s), padding=1)       # [B, cout*s, L]
        got = y.reshape(2, cout, s, 23).permute(0, 1, 3, 2).reshape(2, cout, 23 * s)
        assert torch.allclose(got, ref, atol=1e-5)
        # (phase, channel) column order used by the bf16 path
        wc = conv_transpose_as_conv(w, s).reshape(cout, s, cin, 3).transpose(0, 1).reshape(s * cout, cin, 3)
        y2 = F.conv1d(x, wc, b.repeat(s), padding=1).reshape(2, s, cout, 23).permute(0, 2, 3, 1).reshape(2, cout, 23 * s)
        assert torch.allclose(y2, ref, atol=1e-5)


def test_convt_row_order_groups_phases_by_their_zero_tap():
    """Grouped ConvTranspose rows (include/openvoice_amd.h OV_F_CONVT_GROUPED): a permutation of the natural
    cout*s + phase rows in which every even 32-row tile has an all-zero tap 2 (x[t+1]) and every odd tile an
    all-zero tap 0 (x[t-1]) -- the taps the kernel skips -- with the documented row-in-tile layout."""
    for s, cin, cout in ((8, 6, 16), (2, 6, 64), (8, 512, 256), (2, 64, 32)):
        w = _rand(cin, cout, 2 * s, seed=s)
        wc = conv_transpose_as_conv(w, s)
        order = convt_row_order(cout, s)
        assert sorted(order.tolist()) == list(range(cout * s))
        g = wc[order].reshape(-1, 32, cin, 3)
        assert (g[0::2, :, :, 2] == 0).all() and (g[1::2, :, :, 0] == 0).all()
        assert (g[0::2, :, :, 0] != 0).any() and (g[1::2, :, :, 2] != 0).any()
        for t in range(g.shape[0]):
            q, grp = divmod(t, 2)
            for r in (0, 5, 31):
                co, ph = ((8 * q + r // 4), 4 * grp + r % 4) if s == 8 else ((32 * q + r), grp)
                assert order[32 * t + r] == co * s + ph
    assert convt_row_order(12, 8) is None and convt_row_order(16, 2) is None and convt_row_order(32, 4) is None


def test_gate_row_order_pairs_tanh_and_sigmoid_rows():
    H = 192
    order = gate_row_order(H)
    assert sorted(order.tolist()) == list(range(2 * H))
    for q in range(H // 32):
        blk = order[64 * q: 64 * q + 64]
        assert blk[:32].tolist() == list(range(32 * q, 32 * q + 32))                  # tanh rows of tile pair q
        assert blk[32:].tolist() == list(range(H + 32 * q, H + 32 * q + 32))          # their sigmoid partners


def test_flip_folds_into_channel_order_of_the_coupling_convs():
    """Flip (reference: openvoice/modules.py:374-381) before a coupling == the same coupling with `pre` input
    columns and `post` output rows reversed, applied to the un-flipped tensor with the x0 / x1 halves swapped --
    the identity the engine uses so that no Flip is ever materialised (engine._flow)."""
    C, H, T = 8, 6, 11
    half = C // 2
    x = _rand(2, C, T, seed=1)
    w_pre, b_pre = _rand(H, half, 1, seed=2), _rand(H, seed=3)
    w_post, b_post = _rand(half, H, 1, seed=4), _rand(half, seed=5)

    def coupling(t, wp, bp, wq, bq, x0_first):
        x0, x1 = (t[:, :half], t[:, half:]) if x0_first else (t[:, half:], t[:, :half])
        m = F.conv1d(torch.tanh(F.conv1d(x0, wp, bp)), wq, bq)
        x1n = m + x1
        return torch.cat([x0, x1n], 1) if x0_first else torch.cat([x1n, x0], 1)

    ref = torch.flip(coupling(torch.flip(x, [1]), w_pre, b_pre, w_post, b_post, True), [1])   # flip, couple, flip back
    got = coupling(x, torch.flip(w_pre, [1]), b_pre, torch.flip(w_post, [0]), torch.flip(b_post, [0]), False)
    assert torch.allclose(got, ref, atol=1e-6)


def test_weight_norm_folding_matches_torch():
    v, g = _rand(5, 3, 7, seed=1), _rand(5, 1, 1, seed=2).abs() + 0.1
    sd = {"l.weight_v": v, "l.weight_g": g}
    assert torch.allclose(effective_weight(sd, "l"), torch._weight_norm(v, g, 0), atol=1e-6)
    # ConvTranspose layout: dim 0 is C_in, the norm still runs over every other dim
    vt, gt = _rand(4, 6, 16, seed=3), _rand(4, 1, 1, seed=4).abs() + 0.1
    assert torch.allclose(effective_weight({"u.weight_v": vt, "u.weight_g": gt}, "u"), torch._weight_norm(vt, gt, 0),
                          atol=1e-6)
    assert torch.equal(effective_weight({"p.weight": v}, "p"), v)


def test_padded_frames_is_the_next_multiple_of_four():
    assert [padded_frames(t) for t in (1, 4, 5, 861, 864)] == [4, 4, 8, 864, 864]


def test_spectrogram_dft_weights_reproduce_torch_stft():
    """The DFT-as-conv weights of the native spectrogram (hann * cos / -sin, 4 taps over 256 hop phases) evaluated
    with F.conv1d on CPU equal torch.stft's magnitude: the K = 4 framing-conv formulation itself is exact."""
    import math
    n_fft, hop = 1024, 256
    y = _rand(1, 256 * 9, seed=7) * 0.3
    pad = (n_fft - hop) // 2
    yp = F.pad(y[:, None], (pad, pad), mode="reflect")[:, 0]
    T = (yp.shape[1] - n_fft) // hop + 1
    n = torch.arange(n_fft, dtype=torch.float64)
    win = 0.5 - 0.5 * torch.cos(2 * math.pi * n / n_fft)
    f = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
    re, im = win * torch.cos(2 * math.pi * f * n / n_fft), -win * torch.sin(2 * math.pi * f * n / n_fft)
    hops = yp[:, : (T + 3) * hop].reshape(1, T + 3, hop).transpose(1, 2).double()             # [1, 256, U]
    to_conv = lambda w: w.reshape(-1, n_fft // hop, hop).transpose(1, 2)                        # [rows, 256, 4]
    mag = torch.sqrt(F.conv1d(hops, to_conv(re)) ** 2 + F.conv1d(hops, to_conv(im)) ** 2 + 1e-6)
    ref = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                     return_complex=True).abs().pow(2).add(1e-6).sqrt()
    assert mag.shape == ref.shape and torch.allclose(mag.float(), ref, atol=2e-4)


def test_generator_margin_covers_the_receptive_field_of_the_configuration():
    """``engine.generator_margin_frames``: frames beyond an utterance's end that the generator must still compute under
    ``skip_padding`` -- derived from the checkpoint's configuration instead of the released models' 16.  Checked on the
    CPU oracle's generator (reference: openvoice/models.py:272-291): changing z from frame L + margin on leaves the
    first L frames of audio untouched, for the released configuration and for one with larger ResBlock kernels and
    dilations (where a hard-wired 16 would be too small); three frames less than the released margin is not enough."""
    from openvoice_amd.engine import GENERATOR_MARGIN, generator_margin_frames
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG
    from oracle import vc_oracle
    big = dict(CONVERTER_MODEL_CONFIG, resblock_kernel_sizes=[3, 7, 13], resblock_dilation_sizes=[[1, 3, 7]] * 3)
    assert generator_margin_frames(CONVERTER_MODEL_CONFIG) <= GENERATOR_MARGIN < generator_margin_frames(big)
    L = 6
    for cfg, margins in ((CONVERTER_MODEL_CONFIG, (generator_margin_frames(CONVERTER_MODEL_CONFIG), 11)),
                         (big, (generator_margin_frames(big), GENERATOR_MARGIN))):
        sd = synthetic_state_dict(cfg, 513, seed=5)
        T = L + max(margins) + 4
        z = _rand(1, cfg["inter_channels"], T, seed=7)
        g = 0.3 * _rand(1, cfg["gin_channels"], 1, seed=8)
        with torch.no_grad():
            full = vc_oracle.generator(sd, z, g, cfg)
            hop = full.shape[2] // T
            for m, enough in zip(margins, (True, False)):
                z2 = z.clone()
                z2[:, :, L + m:] = 0.0                                       # what lies beyond the computed frames
                diff = (vc_oracle.generator(sd, z2, g, cfg)[:, :, :L * hop] - full[:, :, :L * hop]).abs().max().item()
                assert (diff == 0.0) == enough, (cfg["resblock_kernel_sizes"], m, diff)


def test_generator_margin_released_configs():
    """The derived bound for the released V1 / V2 converter configurations (the same generator hyper-parameters:
    reference checkpoints' config.json) is 15 frames; the engine rounds it up to GENERATOR_MARGIN = 16."""
    from openvoice_amd.engine import GENERATOR_MARGIN, generator_margin_frames
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG, default_converter_hparams
    assert generator_margin_frames(CONVERTER_MODEL_CONFIG) == 15
    for version in ("v1", "v2"):
        assert generator_margin_frames(dict(default_converter_hparams(version).model.items())) == 15
    assert max(GENERATOR_MARGIN, 15) == 16


def test_validate_config_names_the_unsupported_field():
    """``engine.validate_config``: the reference is config-driven (openvoice/models.py:225-270, ResBlock1 / ResBlock2 at
    :242); the kernels cover the released family and every other value is rejected BY FIELD NAME at load time."""
    import pytest
    from openvoice_amd._lib import OvError
    from openvoice_amd.engine import validate_config
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as base
    validate_config(base)
    validate_config(dict(base, upsample_rates=[8, 8, 4], upsample_kernel_sizes=[16, 16, 8], upsample_initial_channel=256,
                         inter_channels=128, hidden_channels=128, resblock_kernel_sizes=[3, 7],
                         resblock_dilation_sizes=[[1, 3], [1, 5]]))
    cases = [
        (dict(resblock="2"), "resblock", "ResBlock2"),
        (dict(resblock_kernel_sizes=[3, 5, 11]), "resblock_kernel_sizes", "kernel size 5"),
        (dict(resblock_dilation_sizes=[[1, 2, 5]] * 3), "resblock_dilation_sizes", "dilation 2"),
        (dict(upsample_kernel_sizes=[16, 16, 4, 8]), "upsample_kernel_sizes", "2 * stride"),
        (dict(upsample_rates=[8, 8, 3, 2], upsample_kernel_sizes=[16, 16, 6, 4]), "upsample_rates", "divide 32"),
        (dict(upsample_initial_channel=256), "upsample_initial_channel", "16 channels"),
        (dict(hidden_channels=100), "hidden_channels", "multiple of 32"),
        (dict(inter_channels=96), "inter_channels", "multiple of 64"),
    ]
    for change, field, why in cases:
        with pytest.raises(OvError) as e:
            validate_config(dict(base, **change))
        assert repr(field) in str(e.value) and why in str(e.value), (change, str(e.value))
