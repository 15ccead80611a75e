"""Winograd-domain fp32 conv (csrc/conv1d_wino.h, ``ov_conv1d_wino_f32``) against a float64 conv of the same operands
and against the direct fp32 MFMA kernel: every instance, ragged lengths around the 128-column tile, one and several
M-blocks, residual / running-sum / scale operands, padded rows, forced workgroup counts (items that span utterances).
Bar: max-abs error vs float64 <= 16x the direct kernel's (measured ~4x) and <= 2e-5 of the output scale.
reference: openvoice/modules.py:296-309."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _setup(C, K, B, L, seed, cout=None, dil=1):
    from openvoice_amd import wino
    from openvoice_amd.engine import PackedConv
    cout = C if cout is None else cout
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, C, K, generator=gen) * (C * K) ** -0.5
    b = torch.randn(cout, generator=gen) * 0.1
    x = torch.randn(B, C, L, generator=gen).to(DEV)
    return w, b, x, PackedConv(w, b, DEV, K=K, dil=dil), wino.PackedConvWino(w, b, DEV, dil=dil), gen


def _f64(x, w, b, K, slope, dil=1):
    xa = F.leaky_relu(x.double(), slope)
    return F.conv1d(xa.cpu(), w.double(), b.double(), padding=(K - 1) // 2 * dil, dilation=dil).to(x.device)


@pytest.mark.parametrize("frags", [1, 2])
@pytest.mark.parametrize("C,K", [(128, 11), (128, 7), (128, 3), (256, 11), (256, 3), (64, 11), (64, 7), (64, 3), (32, 11), (96, 11)])
@pytest.mark.parametrize("B,L", [(2, 1000), (1, 128), (3, 132), (1, 4), (2, 260), (2, 516), (2, 2060)])
def test_wino_matches_float64_and_direct(C, K, B, L, frags):
    if C < 128 and frags == 1:
        pytest.skip("64- and 32-row layers run two fragments per wave")
    from openvoice_amd import wino
    from openvoice_amd.engine import launch_conv
    w, b, x, direct, wn, _ = _setup(C, K, B, L, seed=C + K + L)
    out_d = torch.empty(B, C, L, device=DEV)
    out_w = torch.full((B, C, L), float("nan"), device=DEV)
    launch_conv(direct, x, 0, C * L, out_d, 0, C * L, B, L, in_slope=0.1)
    wino.launch_conv_wino(wn, x, C * L, out_w, C * L, B, L, in_slope=0.1, frags=frags)
    ref = _f64(x, w, b, K, 0.1)
    e_d = (out_d.double() - ref).abs().max().item()
    e_w = (out_w.double() - ref).abs().max().item()
    assert e_w <= max(16 * e_d, 1e-6), (e_w, e_d)
    assert e_w <= 2e-5 * max(ref.abs().max().item(), 1.0), (e_w, ref.abs().max().item())


@pytest.mark.parametrize("dil", [3, 5])
@pytest.mark.parametrize("C,K", [(128, 11), (128, 7), (128, 3), (256, 11), (256, 7), (64, 11), (64, 7), (64, 3), (32, 11)])
@pytest.mark.parametrize("B,L", [(2, 1000), (1, 252), (3, 244), (1, 4), (2, 512), (2, 2000)])
def test_dilated_wino_matches_float64_and_direct(C, K, B, L, dil):
    from openvoice_amd import wino
    from openvoice_amd.engine import launch_conv
    w, b, x, direct, wn, gen = _setup(C, K, B, L, seed=C + K + L + dil, dil=dil)
    res = torch.randn(B, C, L, generator=gen).to(DEV)
    out_d = torch.empty(B, C, L, device=DEV)
    out_w = torch.full((B, C, L), float("nan"), device=DEV)
    launch_conv(direct, x, 0, C * L, out_d, 0, C * L, B, L, in_slope=0.1, res=res, res_bs=C * L, scale=0.5)
    wino.launch_conv_wino(wn, x, C * L, out_w, C * L, B, L, in_slope=0.1, res=res, res_bs=C * L, scale=0.5)
    ref = (_f64(x, w, b, K, 0.1, dil) + res.double()) * 0.5
    e_d = (out_d.double() - ref).abs().max().item()
    e_w = (out_w.double() - ref).abs().max().item()
    assert e_w <= max(16 * e_d, 1e-6), (e_w, e_d)
    assert e_w <= 2e-5 * max(ref.abs().max().item(), 1.0), (e_w, ref.abs().max().item())


@pytest.mark.parametrize("dil", [3, 5])
def test_dilated_wino_forced_workgroup_counts_and_in_place_running_sum(dil):
    from openvoice_amd import wino
    C, K, B, L = 256, 7, 3, 700
    w, b, x, _, wn, gen = _setup(C, K, B, L, seed=5 + dil, dil=dil)
    acc0 = torch.randn(B, C, L, generator=gen).to(DEV)
    ref = acc0.clone()
    wino.launch_conv_wino(wn, x, C * L, ref, C * L, B, L, in_slope=0.1, add=ref, add_bs=C * L, scale=1.0 / 3)
    want = (_f64(x, w, b, K, 0.1, dil) + acc0.double()) / 3
    assert (ref.double() - want).abs().max().item() <= 2e-5
    for nwg in (1, 5, 16):
        out = acc0.clone()
        wino.launch_conv_wino(wn, x, C * L, out, C * L, B, L, in_slope=0.1, add=out, add_bs=C * L, scale=1.0 / 3, nwg=nwg)
        assert torch.equal(out, ref)


@pytest.mark.parametrize("C,K,dil", [(64, 11, 1), (64, 7, 5), (64, 3, 3), (32, 11, 1), (32, 11, 3), (32, 11, 5)])
def test_wino_64_and_32_row_layer_limits_and_workgroup_counts(C, K, dil):
    """The two- / one-row-fragment workgroups (two / four column sub-blocks): forced workgroup counts and a length-aware
    work list."""
    from openvoice_amd import wino
    B, L = 3, 2100 if C == 64 else 4300
    w, b, x, _, wn, _ = _setup(C, K, B, L, seed=77 + K, dil=dil)
    ref = torch.empty(B, C, L, device=DEV)
    wino.launch_conv_wino(wn, x, C * L, ref, C * L, B, L, in_slope=0.1)
    assert (ref.double() - _f64(x, w, b, K, 0.1, dil)).abs().max().item() <= 2e-5
    for nwg in (1, 7, 16):
        out = torch.full((B, C, L), float("nan"), device=DEV)
        wino.launch_conv_wino(wn, x, C * L, out, C * L, B, L, in_slope=0.1, nwg=nwg)
        assert torch.equal(out, ref)
    ncol = (128 // C) * (256 if dil == 1 else 4 * (64 // dil) * dil)
    lim = torch.tensor([0, ncol + 3, 5000], dtype=torch.int32, device=DEV)
    out = torch.full((B, C, L), float("nan"), device=DEV)
    wino.launch_conv_wino(wn, x, C * L, out, C * L, B, L, in_slope=0.1, col_limit=lim, col_limit_scale=1)
    for bi in range(B):
        done = min(L, (min(L, int(lim[bi])) + ncol - 1) // ncol * ncol)
        assert torch.equal(out[bi, :, :done], ref[bi, :, :done]) and torch.isnan(out[bi, :, done:]).all()


@pytest.mark.parametrize("frags", [1, 2])
@pytest.mark.parametrize("K", [3, 7, 11])
def test_wino_residual_running_sum_scale_and_padded_rows(K, frags):
    """out = (conv + bias + res + add) * scale with x rows 8 floats longer than L and out rows 4 longer; pad columns of
    out stay untouched."""
    from openvoice_amd import wino
    C, B, L, xld, old = 128, 2, 520, 528, 524
    w, b, _, _, wn, gen = _setup(C, K, B, L, seed=K)
    x = torch.randn(B, C, xld, generator=gen).to(DEV)
    res = torch.randn(B, C, old, generator=gen).to(DEV)
    add = torch.randn(B, C, old, generator=gen).to(DEV)
    out = torch.full((B, C, old), 7.0, device=DEV)
    wino.launch_conv_wino(wn, x, C * xld, out, C * old, B, L, in_slope=0.1, scale=1.0 / 3, res=res, res_bs=C * old, add=add,
                          add_bs=C * old, x_ld=xld, out_ld=old, frags=frags)
    ref = (_f64(x[:, :, :L], w, b, K, 0.1) + res[:, :, :L].double() + add[:, :, :L].double()) / 3
    assert (out[:, :, :L].double() - ref).abs().max().item() <= 2e-5
    assert (out[:, :, L:] == 7.0).all()


@pytest.mark.parametrize("frags", [1, 2])
@pytest.mark.parametrize("nwg", [1, 3, 8, 24])
def test_wino_forced_workgroup_counts_are_bit_identical(nwg, frags):
    """Persistent workgroups walk items across utterances and M-blocks; the result does not depend on the split."""
    from openvoice_amd import wino
    C, K, B, L = 256, 7, 3, 640
    w, b, x, _, wn, _ = _setup(C, K, B, L, seed=99)
    ref = torch.empty(B, C, L, device=DEV)
    wino.launch_conv_wino(wn, x, C * L, ref, C * L, B, L, in_slope=0.1, frags=frags)
    out = torch.full((B, C, L), float("nan"), device=DEV)
    wino.launch_conv_wino(wn, x, C * L, out, C * L, B, L, in_slope=0.1, nwg=nwg, frags=frags)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("C,K,dil", [(128, 11, 1), (128, 3, 5), (64, 7, 1), (64, 11, 3), (32, 11, 1)])
def test_wino_activated_hand_over_equals_activating_on_load(C, K, dil):
    """``out_slope``: the first conv of a ResBlock pair stores lrelu(t) and the second stages it with ``in_slope`` = 1 -- bit
    for bit the tensors the pair produces when the second conv activates on load (openvoice/modules.py:298-301); refused
    together with a residual."""
    import torch.nn.functional as F
    from openvoice_amd import _lib, wino
    B, L = 2, 1100
    w, b, x, _, wn, gen = _setup(C, K, B, L, seed=3 * C + K + dil, dil=dil)
    w2 = torch.randn(C, C, K, generator=gen) * (C * K) ** -0.5
    wn2 = wino.PackedConvWino(w2, torch.zeros(C), DEV, dil=1)
    t_raw, t_act = torch.empty(B, C, L, device=DEV), torch.empty(B, C, L, device=DEV)
    wino.launch_conv_wino(wn, x, C * L, t_raw, C * L, B, L, in_slope=0.1)
    wino.launch_conv_wino(wn, x, C * L, t_act, C * L, B, L, in_slope=0.1, out_slope=0.1)
    assert torch.equal(t_act, F.leaky_relu(t_raw, 0.1))
    o_a, o_b = torch.empty(B, C, L, device=DEV), torch.empty(B, C, L, device=DEV)
    wino.launch_conv_wino(wn2, t_raw, C * L, o_a, C * L, B, L, in_slope=0.1, res=x, res_bs=C * L)
    wino.launch_conv_wino(wn2, t_act, C * L, o_b, C * L, B, L, in_slope=1.0, res=x, res_bs=C * L)
    assert torch.equal(o_a, o_b)
    with pytest.raises(_lib.OvError):
        wino.launch_conv_wino(wn, x, C * L, t_act, C * L, B, L, in_slope=0.1, out_slope=0.1, res=x, res_bs=C * L)


def test_wino_refuses_what_it_cannot_run():
    from openvoice_amd import _lib, wino
    w, b, x, _, wn, _ = _setup(128, 11, 1, 256, seed=1)
    out = torch.empty(1, 128, 256, device=DEV)
    with pytest.raises(_lib.OvError):
        wino.launch_conv_wino(wn, x, 128 * 256, x, 128 * 256, 1, 256)            # out aliases x
    with pytest.raises(_lib.OvError):
        wino.launch_conv_wino(wn, x, 128 * 254, out, 128 * 254, 1, 254)          # L % 4
    assert wino.supported(128, 128, 11, 3) and not wino.supported(128, 128, 11, 2) and not wino.supported(32, 32, 7, 1)
    assert wino.supported(32, 32, 11, 1) and not wino.supported(48, 48, 11, 1)


@pytest.mark.parametrize("K,dil", [(11, 1), (7, 3), (3, 5)])
def test_wino_length_aware_work_list(K, dil):
    """col_limit: blocks at or beyond an utterance's limit are neither computed nor written (NaN poison stays), what is
    computed is bit-identical to the full launch; limits 0, 1, mid-block, a block boundary, beyond L."""
    from openvoice_amd import wino
    C, B, L = 128, 6, 1200
    w, b, x, _, wn, _ = _setup(C, K, B, L, seed=31 + K, dil=dil)
    ncol = 256 if dil == 1 else 4 * (64 // dil) * dil
    full = torch.empty(B, C, L, device=DEV)
    wino.launch_conv_wino(wn, x, C * L, full, C * L, B, L, in_slope=0.1)
    limits = torch.tensor([0, 1, ncol + 5, 2 * ncol, L + 50, 300], dtype=torch.int32, device=DEV)
    for scale, lim in ((1, limits), (2, (limits + 1) // 2)):
        out = torch.full((B, C, L), float("nan"), device=DEV)
        wino.launch_conv_wino(wn, x, C * L, out, C * L, B, L, in_slope=0.1, col_limit=lim, col_limit_scale=scale)
        for bi in range(B):
            cols = min(L, int(lim[bi].item()) * scale)
            done = min(L, (cols + ncol - 1) // ncol * ncol)
            assert torch.equal(out[bi, :, :done], full[bi, :, :done]), (bi, cols)
            assert torch.isnan(out[bi, :, done:]).all(), (bi, cols)
