"""Parity of every HIP kernel against plain fp32 PyTorch on CPU (same op, same seeded inputs),
called through the C ABI.  Run on the MI355X box: ``pytest -m gpu``.

Tolerances: fp32 MFMA is an exact fmaf chain, so the only difference from the CPU reference is
summation order; bounds below are ~1e-5 relative to the output scale (stated per test).
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import (PackedConv, conv_transpose_as_conv, convt_row_order, gate_row_order, launch_conv,  # noqa: E402
                                  _ptr)
from openvoice_amd._lib import (EPI_CONVT, EPI_COUPLE, EPI_GATE, EPI_POSTERIOR, EPI_RESSKIP,  # noqa: E402
                                F_CONVT_GROUPED, F_MASK_V, F_OUT2_INIT)

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=gen)


def _close(got, ref, rel=2e-5, what=""):
    err = (got.cpu() - ref).abs().max().item()
    bound = rel * max(1.0, ref.abs().max().item())
    assert err <= bound, f"{what}: max-abs err {err:.3e} > {bound:.3e}"


RESBLOCK_SHAPES = [(c, k, d) for c in (32, 64, 128, 256) for k in (3, 7, 11) for d in (1, 3, 5)]


@pytest.mark.parametrize("c,k,d", RESBLOCK_SHAPES)
def test_resblock_conv_shapes(c, k, d):
    """The 36 (C, k, dilation) convs of the MRF (reference: openvoice/modules.py:222-283) with the
    fused leaky-ReLU prologue and bias + residual epilogue; L is not a multiple of any tile."""
    B, L = 2, 1096 if c >= 128 else 2312
    x, res = _rand(B, c, L, seed=1), _rand(B, c, L, seed=2)
    w, bias = _rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2) + res
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    xd, rd = x.to(DEV), res.to(DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, xd, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=rd, res_bs=c * L)
    _close(out, ref, what=f"C={c} k={k} d={d}")


@pytest.mark.parametrize("c,k,d,B,L,tpw", [(256, 3, 1, 3, 1000, 0), (256, 11, 5, 5, 520, -1), (128, 7, 3, 7, 1096, 2),
                                           (64, 3, 1, 3, 2056, -1), (32, 11, 1, 9, 2312, 3)])
def test_xcd_contiguous_tile_order_is_bit_identical_to_round_robin(c, k, d, B, L, tpw):
    """The XCD-contiguous work order (each XCD = workgroup id % 8 owns a contiguous eighth of the (utterance, time
    tile, M-block) list; launches padded to multiples of 8 workgroups, the surplus exits) visits every tile exactly
    once: same bits as the round-robin order (OV_F_NO_XCD_MAP) and right against the CPU reference -- tile counts that
    are not multiples of 8, several M-blocks per time tile (C = 256), persistent and forced tiles-per-workgroup
    launches."""
    x, res = _rand(B, c, L, seed=21), _rand(B, c, L, seed=22)
    w, bias = _rand(c, c, k, seed=23, scale=(c * k) ** -0.5), _rand(c, seed=24, scale=0.1)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2) + res
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    xd, rd = x.to(DEV), res.to(DEV)
    outs = []
    for flags in (0, _lib.F_NO_XCD_MAP):
        out = torch.full((B, c, L), float("nan"), device=DEV)
        launch_conv(layer, xd, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=rd, res_bs=c * L, flags=flags,
                    tiles_per_wg=tpw)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], ref, what=f"C={c} k={k} d={d} B={B} L={L} tpw={tpw}")


@pytest.mark.parametrize("c,k,d,L,tpw", [(32, 3, 1, 2312, 2), (32, 11, 5, 2312, 4), (64, 7, 3, 2056, 3),
                                         (128, 3, 5, 1096, 4), (256, 11, 1, 520, 2), (128, 7, 1, 1096, 16)])
def test_resblock_conv_several_tiles_per_workgroup(c, k, d, L, tpw):
    """The loader wave runs one chunk ahead across time-tile boundaries when a workgroup walks
    several tiles (the launcher does this by itself only for grids of >= 8192 tiles): tile counts
    that do not divide evenly, last tile ragged, more tiles per workgroup than exist."""
    B = 2
    x, res, add = _rand(B, c, L, seed=1), _rand(B, c, L, seed=2), _rand(B, c, L, seed=9)
    w, bias = _rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1)
    ref = (F.conv1d(F.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2) + res + add) / 3.0
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=res.to(DEV), res_bs=c * L,
                add=add.to(DEV), add_bs=c * L, scale=1.0 / 3.0, tiles_per_wg=tpw)
    _close(out, ref, what=f"C={c} k={k} d={d} tpw={tpw}")


def test_wavenet_gate_several_tiles_per_workgroup():
    B, H, T = 2, 192, 700
    x, g = _rand(B, H, T, seed=1), _rand(B, 2 * H, seed=2, scale=0.3)
    w_in, b_in = _rand(2 * H, H, 5, seed=3, scale=(5 * H) ** -0.5), _rand(2 * H, seed=4, scale=0.1)
    x_in = F.conv1d(x, w_in, b_in, padding=2) + g[:, :, None]
    acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
    order = gate_row_order(H)
    l_in = PackedConv(w_in[order], b_in[order], DEV, K=5, cout=H)
    actsd = torch.full((B, H, T), float("nan"), device=DEV)
    launch_conv(l_in, x.to(DEV), 0, H * T, actsd, 0, H * T, B, T, epi=EPI_GATE, bias_b=g[:, order].contiguous().to(DEV),
                bias_b_bs=2 * H, rows=2 * H, tiles_per_wg=3)
    _close(actsd, acts, what="gate tpw=3")


@pytest.mark.parametrize("cin,cout,k,L", [(513, 192, 1, 861), (192, 512, 7, 861), (96, 192, 1, 65),
                                          (192, 96, 1, 17), (192, 384, 1, 1), (192, 512, 7, 864)])
def test_frame_rate_convs_unaligned_and_aligned(cin, cout, k, L):
    """1x1 / k7 convs at frame rate: T = 861 takes the 4-byte staging path, 864 the 16-byte one;
    also tiny T (1, 17) and C_in that is not a multiple of the channel chunk (513, 96)."""
    B = 3
    x = _rand(B, cin, L, seed=5)
    w, bias = _rand(cout, cin, k, seed=6, scale=(cin * k) ** -0.5), _rand(cout, seed=7, scale=0.1)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, max(1, L // 2), max(1, L // 3)])[:, None]).float()
    ref = F.conv1d(x, w, bias, padding=(k - 1) // 2) * mask[:, None]
    layer = PackedConv(w, bias, DEV, K=k)
    out = torch.full((B, cout, L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, cin * L, out, 0, cout * L, B, L, flags=F_MASK_V, mask=mask.to(DEV))
    _close(out, ref, what=f"{cin}->{cout} k={k} L={L}")


def test_linear_epilogue_add_scale_and_batch_bias():
    """conv2 of the last ResBlock pair: (conv + bias + res + running MRF sum) / 3, and conv_pre's
    per-utterance conditioning bias (reference: openvoice/models.py:273-286)."""
    B, C, L, k = 2, 64, 520, 3
    x, res, add = _rand(B, C, L, seed=1), _rand(B, C, L, seed=2), _rand(B, C, L, seed=3)
    w, bias, bb = _rand(C, C, k, seed=4, scale=0.1), _rand(C, seed=5), _rand(B, C, seed=6)
    ref = (F.conv1d(x, w, bias, padding=1) + bb[:, :, None] + res + add) * (1.0 / 3.0)
    layer = PackedConv(w, bias, DEV, K=k)
    out = torch.empty(B, C, L, device=DEV)
    bbd = torch.zeros(B, 128)
    bbd[:, :C] = bb
    launch_conv(layer, x.to(DEV), 0, C * L, out, 0, C * L, B, L, res=res.to(DEV), res_bs=C * L, add=add.to(DEV),
                add_bs=C * L, scale=1.0 / 3.0, bias_b=bbd.to(DEV), bias_b_bs=128)
    _close(out, ref, what="linear epilogue")


@pytest.mark.parametrize("T", [1, 17, 200, 861])
def test_wavenet_layer_gate_and_res_skip(T):
    """One WN layer = k5 conv + conditioning + tanh*sigmoid gate, then the 1x1 res/skip conv
    (reference: openvoice/modules.py:192-209, openvoice/commons.py:100-107)."""
    B, H = 2, 192
    x, g = _rand(B, H, T, seed=1), _rand(B, 2 * H, seed=2, scale=0.3)
    w_in, b_in = _rand(2 * H, H, 5, seed=3, scale=(5 * H) ** -0.5), _rand(2 * H, seed=4, scale=0.1)
    w_rs, b_rs = _rand(2 * H, H, 1, seed=5, scale=H ** -0.5), _rand(2 * H, seed=6, scale=0.1)
    skip0 = _rand(B, H, T, seed=7)
    mask = (torch.arange(T)[None, :] < torch.tensor([T, max(1, T - 5)])[:, None]).float()
    x_in = F.conv1d(x, w_in, b_in, padding=2) + g[:, :, None]
    acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    x_ref = (x + rs[:, :H]) * mask[:, None]
    skip_ref = skip0 + rs[:, H:]
    order = gate_row_order(H)
    l_in = PackedConv(w_in[order], b_in[order], DEV, K=5, cout=H)
    l_rs = PackedConv(w_rs, b_rs, DEV, K=1)
    xd, maskd = x.to(DEV), mask.to(DEV)
    actsd = torch.full((B, H, T), float("nan"), device=DEV)
    gd = g[:, order].contiguous().to(DEV)
    launch_conv(l_in, xd, 0, H * T, actsd, 0, H * T, B, T, epi=EPI_GATE, bias_b=gd, bias_b_bs=2 * H, rows=2 * H)
    _close(actsd, acts, what="gate")
    skipd = skip0.to(DEV)
    launch_conv(l_rs, actsd, 0, H * T, xd, 0, H * T, B, T, epi=EPI_RESSKIP, out2=skipd, out2_bs=H * T, mask=maskd,
                split=H)
    _close(xd, x_ref, what="residual")
    _close(skipd, skip_ref, what="skip +=")
    # first-layer form (skip = ...) and last-layer form (all rows are skip)
    skipd2 = torch.full((B, H, T), float("nan"), device=DEV)
    l_last = PackedConv(w_rs[:H], b_rs[:H], DEV, K=1)
    launch_conv(l_last, actsd, 0, H * T, xd, 0, H * T, B, T, epi=EPI_RESSKIP, flags=F_OUT2_INIT, out2=skipd2,
                out2_bs=H * T, mask=maskd, split=0)
    _close(skipd2, F.conv1d(acts, w_rs[:H], b_rs[:H]), what="skip = (last layer)")


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("flipped", [False, True])
def test_coupling_combine_with_folded_flip(reverse, flipped):
    """post 1x1 conv + mean-only coupling update written in place into the x1 half, with the
    Flip folded into channel order (reference: openvoice/modules.py:441-455, :374-381)."""
    B, H, C, T = 2, 192, 192, 77
    half = C // 2
    h, x = _rand(B, H, T, seed=1), _rand(B, C, T, seed=2)
    w, b = _rand(half, H, 1, seed=3, scale=H ** -0.5), _rand(half, seed=4, scale=0.1)
    mask = (torch.arange(T)[None, :] < torch.tensor([T, 40])[:, None]).float()[:, None]
    # logical view: when flipped, the logical tensor is the channel-reversed physical one
    logical = torch.flip(x, [1]) if flipped else x
    m = F.conv1d(h, w, b) * mask
    x1 = logical[:, half:]
    x1n = (x1 - m) * mask if reverse else m + x1 * mask
    logical_new = torch.cat([logical[:, :half], x1n], 1)
    ref = torch.flip(logical_new, [1]) if flipped else logical_new
    wl, bl = (torch.flip(w, [0]), torch.flip(b, [0])) if flipped else (w, b)
    layer = PackedConv(wl, bl, DEV, K=1)
    xd = x.to(DEV)
    launch_conv(layer, h.to(DEV), 0, H * T, xd, 0 if flipped else half * T, C * T, B, T, epi=EPI_COUPLE,
                mask=mask[:, 0].contiguous().to(DEV), scale=-1.0 if reverse else 1.0)
    _close(xd, ref, what=f"couple reverse={reverse} flipped={flipped}")


def test_posterior_sample_epilogue():
    """proj 1x1 conv + z = (m + noise*tau*exp(logs)) * mask (reference: openvoice/models.py:218-220)."""
    B, H, C, T, tau = 2, 192, 192, 130, 0.3
    h, noise = _rand(B, H, T, seed=1), _rand(B, C, T, seed=2)
    w, b = _rand(2 * C, H, 1, seed=3, scale=H ** -0.5), _rand(2 * C, seed=4, scale=0.1)
    mask = (torch.arange(T)[None, :] < torch.tensor([T, 64])[:, None]).float()[:, None]
    stats = F.conv1d(h, w, b) * mask
    ref = (stats[:, :C] + noise * tau * torch.exp(stats[:, C:])) * mask
    order = gate_row_order(C)
    layer = PackedConv(w[order], b[order], DEV, K=1, cout=C)
    z = torch.full((B, C, T), float("nan"), device=DEV)
    launch_conv(layer, h.to(DEV), 0, H * T, z, 0, C * T, B, T, epi=EPI_POSTERIOR, res=noise.to(DEV), res_bs=C * T,
                scale=tau, mask=mask[:, 0].contiguous().to(DEV), rows=2 * C)
    _close(z, ref, what="posterior")


@pytest.mark.parametrize("cin,cout,s,L", [(512, 256, 8, 61), (256, 128, 8, 488), (128, 64, 2, 1000), (64, 32, 2, 2000)])
def test_conv_transpose_as_phase_conv(cin, cout, s, L):
    """leaky_relu(0.1) + ConvTranspose1d(k=2s, stride s, pad s/2) (reference: openvoice/models.py:278-279,
    :244-256) as the 3-tap phase conv with the interleaving 16/8-byte store epilogue."""
    B, k = 2, 2 * s
    x = _rand(B, cin, L, seed=1)
    w, b = _rand(cin, cout, k, seed=2, scale=(2 * cin) ** -0.5), _rand(cout, seed=3, scale=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2)
    assert ref.shape[2] == s * L
    wc, bc = conv_transpose_as_conv(w, s), b.repeat_interleave(s)
    # natural row order (cout*s + phase): the generic kernel, all three taps
    layer = PackedConv(wc, bc, DEV, K=3, cout=cout)
    out = torch.full((B, cout, s * L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, cin * L, out, 0, cout * s * L, B, L, epi=EPI_CONVT, in_slope=0.1, phase_s=s)
    _close(out, ref, what=f"convT {cin}->{cout} s={s}")
    # grouped row order (what the engine uses): the all-zero tap of each phase group is skipped
    order = convt_row_order(cout, s)
    assert order is not None
    layer = PackedConv(wc[order], bc[order], DEV, K=3, cout=cout)
    out2 = torch.full((B, cout, s * L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, cin * L, out2, 0, cout * s * L, B, L, epi=EPI_CONVT, in_slope=0.1, phase_s=s,
                flags=F_CONVT_GROUPED)
    _close(out2, ref, what=f"grouped convT {cin}->{cout} s={s}")
    assert torch.equal(out, out2)    # skipping a zero tap changes no fmaf


@pytest.mark.parametrize("L", [4352, 1001])
def test_conv_post_tanh(L):
    """leaky_relu(0.01) + conv k7 32->1 (no bias) + tanh (reference: openvoice/models.py:287-289)."""
    B, C = 2, 32
    x, w = _rand(B, C, L, seed=1), _rand(1, C, 7, seed=2, scale=0.1)
    ref = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), w, None, padding=3))
    lib = _lib.load()
    out = torch.full((B, 1, L), float("nan"), device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, wd = x.to(DEV), w[0].contiguous().to(DEV)
    assert lib.ov_conv_post_tanh_f32(_ptr(xd), _ptr(wd), _ptr(out), B, C, L, 7, 0.01, st) == 0
    _close(out, ref, what="conv_post")


def test_linear_and_sequence_mask():
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x, w, b = _rand(3, 256, seed=1), _rand(1536, 256, seed=2, scale=1 / 16), _rand(1536, seed=3)
    y = torch.empty(3, 1536, device=DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    assert lib.ov_linear_f32(_ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), 3, 1536, 256, st) == 0
    _close(y, x @ w.t() + b, what="linear")
    lengths = torch.tensor([5, 0, 9, 7], dtype=torch.int64, device=DEV)
    mask = torch.empty(4, 9, device=DEV)
    assert lib.ov_sequence_mask_f32(ctypes.c_void_p(lengths.data_ptr()), _ptr(mask), 4, 9, 0, st) == 0
    ref = (torch.arange(9)[None] < lengths.cpu()[:, None]).float()
    assert torch.equal(mask.cpu(), ref)


def _pad_rows(t, ld, fill=float("nan")):
    """[B, C, L] -> [B, C, ld] with the pad columns poisoned: a kernel that reads them as data or
    writes them fails the comparison."""
    B, C, L = t.shape
    out = torch.full((B, C, ld), fill)
    out[:, :, :L] = t
    return out


@pytest.mark.parametrize("L", [861, 862, 863, 17, 1])
def test_padded_rows_ragged_length_16_byte_staging(L):
    """Frame-rate tensors live in rows padded to a multiple of 4 floats (engine.padded_frames) so the
    16-byte staging path is taken for any T; the valid length stays ragged.  One WN layer + the k7
    conv_pre shape, pad columns poisoned with NaN on the way in and checked untouched on the way out."""
    B, H, ld = 2, 192, (L + 3) // 4 * 4 + 4
    x, g = _rand(B, H, L, seed=1), _rand(B, 2 * H, seed=2, scale=0.3)
    w_in, b_in = _rand(2 * H, H, 5, seed=3, scale=(5 * H) ** -0.5), _rand(2 * H, seed=4, scale=0.1)
    w_rs, b_rs = _rand(2 * H, H, 1, seed=5, scale=H ** -0.5), _rand(2 * H, seed=6, scale=0.1)
    skip0 = _rand(B, H, L, seed=7)
    lens = torch.tensor([L, max(1, L - 5)])
    mask = (torch.arange(L)[None, :] < lens[:, None]).float()
    x_in = F.conv1d(x, w_in, b_in, padding=2) + g[:, :, None]
    acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    x_ref, skip_ref = (x + rs[:, :H]) * mask[:, None], skip0 + rs[:, H:]
    order = gate_row_order(H)
    l_in = PackedConv(w_in[order], b_in[order], DEV, K=5, cout=H)
    l_rs = PackedConv(w_rs, b_rs, DEV, K=1)
    xd = _pad_rows(x, ld).to(DEV)
    maskd = _pad_rows(mask[:, None], ld)[:, 0].contiguous().to(DEV)
    actsd = torch.full((B, H, ld), float("nan"), device=DEV)
    gd = g[:, order].contiguous().to(DEV)
    launch_conv(l_in, xd, 0, H * ld, actsd, 0, H * ld, B, L, epi=EPI_GATE, bias_b=gd, bias_b_bs=2 * H, rows=2 * H,
                x_ld=ld, out_ld=ld)
    _close(actsd[:, :, :L], acts, what="gate (padded rows)")
    assert torch.isnan(actsd[:, :, L:]).all(), "gate wrote into the pad columns"
    skipd = _pad_rows(skip0, ld).to(DEV)
    launch_conv(l_rs, actsd, 0, H * ld, xd, 0, H * ld, B, L, epi=EPI_RESSKIP, out2=skipd, out2_bs=H * ld, mask=maskd,
                mask_bs=ld, split=H, x_ld=ld, out_ld=ld)
    _close(xd[:, :, :L], x_ref, what="residual (padded rows)")
    _close(skipd[:, :, :L], skip_ref, what="skip (padded rows)")
    assert torch.isnan(xd[:, :, L:]).all() and torch.isnan(skipd[:, :, L:]).all()
    # conv_pre shape: k7 192 -> 512 with a per-utterance bias, padded in, padded out
    w7, b7 = _rand(512, H, 7, seed=8, scale=(7 * H) ** -0.5), _rand(512, seed=9, scale=0.1)
    l7 = PackedConv(w7, b7, DEV, K=7)
    out = torch.full((B, 512, ld), float("nan"), device=DEV)
    launch_conv(l7, _pad_rows(x, ld).to(DEV), 0, H * ld, out, 0, 512 * ld, B, L, x_ld=ld, out_ld=ld)
    _close(out[:, :, :L], F.conv1d(x, w7, b7, padding=3), what="k7 (padded rows)")
    assert torch.isnan(out[:, :, L:]).all()


@pytest.mark.parametrize("s,cin,cout,L", [(8, 512, 256, 61), (2, 128, 64, 999)])
def test_conv_transpose_from_padded_rows(s, cin, cout, L):
    """ups.0 reads the padded frame-rate rows (x_ld > L) and writes dense upsampled rows."""
    B, k, ld = 2, 2 * s, (L + 3) // 4 * 4
    x = _rand(B, cin, L, seed=1)
    w, b = _rand(cin, cout, k, seed=2, scale=(2 * cin) ** -0.5), _rand(cout, seed=3, scale=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2)
    order = convt_row_order(cout, s)
    layer = PackedConv(conv_transpose_as_conv(w, s)[order], b.repeat_interleave(s)[order], DEV, K=3, cout=cout)
    out = torch.full((B, cout, s * L), float("nan"), device=DEV)
    launch_conv(layer, _pad_rows(x, ld).to(DEV), 0, cin * ld, out, 0, cout * s * L, B, L, epi=EPI_CONVT,
                in_slope=0.1, phase_s=s, x_ld=ld, flags=F_CONVT_GROUPED)
    _close(out, ref, what=f"convT from padded rows s={s}")


@pytest.mark.parametrize("c,tile,loaders,chunk", [(32, 3, 4, 16), (32, 4, 4, 16), (64, 2, 4, 32), (64, 2, 4, 16),
                                                  (128, 1, 2, 32), (128, 1, 2, 16), (256, 1, 2, 32),
                                                  (32, 1, 2, 32), (64, 1, 2, 16)])
@pytest.mark.parametrize("k,d", [(3, 1), (7, 3), (11, 5)])
def test_every_tile_chunk_and_loader_count(c, tile, loaders, chunk, k, d):
    """All (tile, channels-per-chunk, loader-wave count) instantiations of the MRF convs give the same
    answer: tile ids 1..4 = 128x128, 64x256, 32x512, 32x256; forcing one must not fall back silently."""
    B, L = 2, 2312
    x, res = _rand(B, c, L, seed=1), _rand(B, c, L, seed=2)
    w, bias = _rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2) + res
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=res.to(DEV), res_bs=c * L,
                tile=tile, loaders=loaders, chunk=chunk)
    _close(out, ref, what=f"C={c} k={k} d={d} tile={tile} loaders={loaders} chunk={chunk}")


def test_forced_variant_that_does_not_exist_is_an_error():
    c, k, L, B = 32, 5, 256, 1
    layer = PackedConv(_rand(c, c, k, seed=1), None, DEV, K=k)
    x, out = torch.zeros(B, c, L, device=DEV), torch.zeros(B, c, L, device=DEV)
    with pytest.raises(_lib.OvError, match="OV_E_UNSUPPORTED"):
        launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, tile=3)   # no k=5 32x512 instantiation


def test_residual_is_preloaded_only_without_value_mask():
    """LINEAR with a residual starts the accumulators at res (+ add); with OV_F_MASK_V the reference
    order v*mask + res must be kept (the mask must not touch the residual)."""
    B, C, L, k = 2, 64, 777, 3
    x, res, add = _rand(B, C, L, seed=1), _rand(B, C, L, seed=2), _rand(B, C, L, seed=3)
    w, bias = _rand(C, C, k, seed=4, scale=0.1), _rand(C, seed=5)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, 300])[:, None]).float()
    conv = F.conv1d(x, w, bias, padding=1)
    layer = PackedConv(w, bias, DEV, K=k)
    out = torch.full((B, C, L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, C * L, out, 0, C * L, B, L, res=res.to(DEV), res_bs=C * L, add=add.to(DEV),
                add_bs=C * L, scale=0.5, flags=F_MASK_V, mask=mask.to(DEV))
    _close(out, (conv * mask[:, None] + res + add) * 0.5, what="mask_v + res + add")
    launch_conv(layer, x.to(DEV), 0, C * L, out, 0, C * L, B, L, res=res.to(DEV), res_bs=C * L, scale=0.5)
    _close(out, (conv + res) * 0.5, what="res only")


@pytest.mark.parametrize("B,N", [(2, 256 * 17), (1, 220500), (3, 5000), (1, 385)])
def test_native_spectrogram_matches_torch_stft(B, N):
    """ov_frame_hops_f32 + the K = 4 framing conv with the magnitude epilogue against the reference
    definition (openvoice/mel_processing.py:40-75) evaluated with torch.stft on the CPU.  |spec| reaches
    ~10^2 (a 1024-sample window of a full-scale sinusoid), the bound is 2e-5 of that."""
    from openvoice_amd.mel_processing import spectrogram_torch
    gen = torch.Generator().manual_seed(N)
    t = torch.arange(N, dtype=torch.float32) / 22050
    y = 0.6 * torch.sin(2 * torch.pi * (200 + 300 * torch.arange(B)[:, None]) * t) + 0.05 * torch.randn(B, N, generator=gen)
    ref = spectrogram_torch(y, 1024, 22050, 256, 1024, center=False)                 # CPU: torch.stft
    got = spectrogram_torch(y.to(DEV), 1024, 22050, 256, 1024, center=False)         # device: native kernels
    assert got.shape == ref.shape == (B, 513, (N - 256) // 256 + 1)
    _close(got, ref, what=f"spectrogram B={B} N={N}")
    # the view's storage rows are 16-byte aligned and padded columns hold no garbage
    assert got.stride(2) == 1 and got.stride(1) % 4 == 0


def test_spectrogram_on_device_never_falls_back_to_rocfft():
    """A device tensor with a configuration the native kernels do not cover must raise, not take torch.stft."""
    from openvoice_amd.mel_processing import spectrogram_torch
    y = torch.zeros(1, 4096, device=DEV)
    for kw in (dict(n_fft=1024, hop_size=200, win_size=1024), dict(n_fft=1024, hop_size=256, win_size=800),
               dict(n_fft=2048, hop_size=256, win_size=2048)):
        with pytest.raises(_lib.OvError):
            spectrogram_torch(y, kw["n_fft"], 22050, kw["hop_size"], kw["win_size"], center=False)
    with pytest.raises(_lib.OvError):
        spectrogram_torch(y, 1024, 22050, 256, 1024, center=True)
    with pytest.raises(_lib.OvError):
        spectrogram_torch(y.double(), 1024, 22050, 256, 1024, center=False)
