"""Split-precision Conv1d (ov_conv1d_split3, csrc/conv1d_split3.h; VERDICT r04 item 1: fp32-level products on the bf16
matrix pipe, opt-in) against a float64 F.conv1d of the SAME fp32 operands: every (C, K, dilation) with an instance,
lengths around the 128-row step, utterance edges, forced workgroup counts (ranges that start mid-utterance), the residual
form with an activated residual, scale / output activation, both product counts; and the two layout kernels.
Bars: 6 products = fp32 level (1e-5 of scale: the fp32 MFMA kernels measure 2-4e-6 on these shapes); 3 products =
16-bit operands (2e-4).  reference: openvoice/modules.py:296-309."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd.split3 import (PackedConvSplit3, from_planes, launch_conv_split3, split3_reference, supported,  # noqa: E402
                                  to_planes)

DEV = "cuda:0"
SLOPE = 0.1
KD = [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)]


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _layer(c, k, d, seed=0):
    w, b = _rand(c, c, k, seed=seed + 1, scale=(c * k) ** -0.5), _rand(c, seed=seed + 2, scale=0.1)
    return w, b, PackedConvSplit3(w, b, DEV, dil=d)


def _value(planes):
    p = planes.float()
    return ((p[0] + p[1]) + p[2]).transpose(1, 2)          # (B, C, L)


def _reference(x, w, b, k, d, res=None, scale=1.0, out_slope=1.0):
    """x (B, C, L) fp32 raw; the kernel reads lrelu(x) (stored activated) and adds the RAW residual."""
    xa = F.leaky_relu(x, SLOPE).double()
    y = F.conv1d(xa, w.double(), b.double(), dilation=d, padding=(k - 1) * d // 2)
    if res is not None:
        y = y + res.double()
    return F.leaky_relu(y * scale, out_slope)


def _run(x, layer, res_planes=None, res_slope=1.0, **kw):
    xp = to_planes(x.to(DEV), SLOPE)
    out = torch.full((3,) + tuple(xp.shape[1:3]) + (layer.cout,), float("nan"), dtype=torch.bfloat16, device=DEV)
    launch_conv_split3(layer, xp, out, res=res_planes, res_slope=res_slope, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all(), "unwritten (NaN-poisoned) output elements"
    return out


def _check(out, ref, bar):
    err = (_value(out).double().cpu() - ref).abs().max().item()
    assert err <= bar * max(1.0, ref.abs().max().item()), (err, bar)
    return err


def test_layout_kernels_round_trip_exactly():
    x = _rand(3, 128, 333, seed=3) * torch.exp(3 * _rand(3, 128, 333, seed=4))
    xp = to_planes(x.to(DEV), SLOPE)
    want = split3_reference(F.leaky_relu(x, SLOPE).transpose(1, 2).contiguous())
    assert torch.equal(xp.cpu(), want)
    back = from_planes(xp, in_slope=1.0)
    assert torch.equal(back.cpu(), F.leaky_relu(x, SLOPE))
    raw = to_planes(x.to(DEV), 1.0)
    both = from_planes(raw, raw, None, scale=0.5)
    assert torch.equal(both.cpu(), (x + x) * 0.5)
    inv = from_planes(xp, in_slope=SLOPE).cpu()              # the inverse activation: exact to one rounding
    assert (inv - x).abs().max().item() <= 2.0 ** -22 * x.abs().max().item()


@pytest.mark.parametrize("c", [64, 128, 256])
@pytest.mark.parametrize("k,d", KD)
def test_split3_conv_matches_float64(c, k, d):
    assert supported(c, c, k, d)
    B, L = 2, 700
    w, b, layer = _layer(c, k, d)
    x = _rand(B, c, L, seed=9)
    err = _check(_run(x, layer), _reference(x, w, b, k, d), 1e-5)
    print(f"C={c} K={k} d={d}: max-abs vs float64 {err:.2e}")


@pytest.mark.parametrize("L", [1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1030])
@pytest.mark.parametrize("c,k,d", [(128, 11, 5), (128, 3, 1), (256, 7, 3), (64, 11, 5), (64, 3, 1)])
def test_split3_lengths_around_the_step(c, k, d, L):
    B = 3
    w, b, layer = _layer(c, k, d, seed=L)
    x = _rand(B, c, L, seed=L + 5)
    _check(_run(x, layer), _reference(x, w, b, k, d), 1e-5)


@pytest.mark.parametrize("nwg", [1, 2, 3, 7, 1000])
def test_split3_forced_workgroup_counts(nwg):
    """Ranges of steps that start / end mid-utterance and span utterances; C = 256 also splits the channel blocks."""
    for c, k, d, L in [(128, 7, 1, 530), (256, 3, 3, 390), (64, 7, 5, 530)]:
        w, b, layer = _layer(c, k, d, seed=nwg)
        x = _rand(3, c, L, seed=nwg + 1)
        _check(_run(x, layer, nwg=nwg), _reference(x, w, b, k, d), 1e-5)


@pytest.mark.parametrize("c,k", [(128, 3), (128, 11), (256, 7), (64, 3), (64, 11)])
@pytest.mark.parametrize("L", [64, 200, 515])
def test_split3_residual_form(c, k, L):
    """conv2 of a ResBlock pair (modules.py:301-306): the residual is the pair's input, stored activated; the output is
    stored activated for the next pair, or scaled (the MRF mean)."""
    B = 2
    w, b, layer = _layer(c, k, 1, seed=L)
    t, xres = _rand(B, c, L, seed=L + 1), _rand(B, c, L, seed=L + 2)
    res_planes = to_planes(xres.to(DEV), SLOPE)              # as the pair's input is stored
    ref = _reference(t, w, b, k, 1, res=xres, out_slope=SLOPE)
    _check(_run(t, layer, res_planes=res_planes, res_slope=SLOPE, out_slope=SLOPE), ref, 1e-5)
    ref = _reference(t, w, b, k, 1, res=xres, scale=1.0 / 3.0)
    raw = to_planes(xres.to(DEV), 1.0)
    _check(_run(t, layer, res_planes=raw, res_slope=1.0, scale=1.0 / 3.0), ref, 1e-5)


@pytest.mark.parametrize("c", [64, 128])
def test_split3_three_products_is_the_16_bit_mode(c):
    k, d = 11, 1
    w, b, layer = _layer(c, k, d)
    x = _rand(2, c, 400, seed=5)
    ref = _reference(x, w, b, k, d)
    e3 = _check(_run(x, layer, products=3), ref, 2e-4)
    e6 = _check(_run(x, layer, products=6), ref, 1e-5)
    print(f"3 products {e3:.2e}, 6 products {e6:.2e}")
    assert e6 < e3


def test_split3_wide_dynamic_range():
    """Operands spanning 2^+-12: every plane product keeps its own exponent (bf16 has fp32's range), so the error bar is
    relative to the output scale as for fp32."""
    c, k, d = 128, 7, 3
    w, b, layer = _layer(c, k, d)
    x = _rand(2, c, 300, seed=6) * torch.exp2(4 * _rand(2, c, 300, seed=7).clamp(-3, 3))
    _check(_run(x, layer), _reference(x, w, b, k, d), 1e-5)


def test_split3_rejects_what_it_has_no_instance_for():
    from openvoice_amd._lib import OvError
    w, b, layer = _layer(128, 3, 3)
    x = to_planes(_rand(1, 128, 64).to(DEV), SLOPE)
    out = torch.empty_like(x)
    with pytest.raises(OvError):
        launch_conv_split3(layer, x, out, res=x, res_slope=SLOPE)          # residual form: dilation 1 only / aliasing
    with pytest.raises(OvError):
        launch_conv_split3(layer, x, out, products=4)
    with pytest.raises(OvError):
        PackedConvSplit3(_rand(32, 32, 3), None, DEV)


@pytest.mark.parametrize("c,k,d,res", [(128, 7, 3, False), (256, 3, 1, True), (64, 11, 1, True)])
def test_split3_length_aware_work_list(c, k, d, res):
    """``col_limit``: 128-row time tiles that start at or beyond an utterance's limit are neither computed nor written (the
    output stays NaN-poisoned there); everything that IS computed is bit-identical to the launch without limits; the dense
    work list is re-dealt over the workgroups whatever the lengths are (limits 0, 1, mid-tile, a tile boundary, > L)."""
    B, L = 6, 700
    w, b, layer = _layer(c, k, d, seed=3)
    x = _rand(B, c, L, seed=4)
    xp = to_planes(x.to(DEV), SLOPE)
    resp = to_planes(_rand(B, c, L, seed=5).to(DEV), SLOPE) if res else None
    kw = dict(res=resp, res_slope=SLOPE) if res else {}
    full = torch.empty_like(xp)
    launch_conv_split3(layer, xp, full, **kw)
    limits = torch.tensor([0, 1, 300, 128, 9999, 513], dtype=torch.int32, device=DEV)
    for scale, nwg in ((1, 0), (1, 3), (2, 0)):
        out = torch.full_like(xp, float("nan"))
        launch_conv_split3(layer, xp, out, col_limit=limits, col_limit_scale=scale, nwg=nwg, **kw)
        torch.cuda.synchronize()
        for bi, lim in enumerate(limits.tolist()):
            cols = min(L, lim * scale)
            kept = min(L, (cols + 127) // 128 * 128)              # whole tiles up to the limit
            assert torch.equal(out[:, bi, :kept], full[:, bi, :kept]), (bi, scale, nwg)
            assert torch.isnan(out[:, bi, kept:].float()).all(), (bi, scale, nwg)
    with pytest.raises(Exception):
        launch_conv_split3(layer, xp, full, col_limit=limits, col_limit_scale=0, **kw)


FULL = [(256, 11, 1, 6888), (128, 11, 5, 55104), (128, 3, 1, 55104), (64, 7, 3, 110208)]


@pytest.mark.parametrize("c,k,d,L", FULL)
def test_split3_full_size_properties(c, k, d, L):
    """BASELINE's full sizes (32 utterances, the MRF stages' real lengths), where a float64 reference is out of reach: three
    size-independent properties, each BIT-EXACT because an output element's summation order does not depend on where it sits.
    (a) time-shift equivariance -- the input moved by 37 columns (not a multiple of any tile or chunk size) gives the output
    moved by 37 columns away from the utterance edges: every tile's halo rows and every work-list boundary are read right;
    (b) utterance independence -- a permuted batch gives the permuted output; (c) scaling by a power of two (bias-free,
    linear epilogue) scales every plane exactly."""
    B, s = 32, 37
    w, _, _ = _layer(c, k, d, seed=c + k)
    layer = PackedConvSplit3(w, None, DEV, dil=d)
    gen = torch.Generator(device=DEV).manual_seed(L + k)
    x = torch.randn(B, L + s, c, device=DEV, generator=gen) * torch.exp(torch.randn(B, 1, c, device=DEV, generator=gen))

    def planes(v):                                           # (B, L, C) fp32 -> (3, B, L, C) bf16, exact split
        return split3_reference(v.contiguous()).contiguous()

    def run(xp):
        out = torch.full((3, B, L, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        launch_conv_split3(layer, xp, out)
        return out

    base = run(planes(x[:, s:]))                             # columns s .. L + s
    torch.cuda.synchronize()
    assert torch.isfinite(base.float()).all()
    moved = run(planes(x[:, :L]))                            # columns 0 .. L: the same signal s columns later
    halo = (k - 1) * d // 2
    assert torch.equal(moved[:, :, s + halo:L - halo], base[:, :, halo:L - s - halo])
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    assert torch.equal(run(planes(x[perm, s:])), base[:, perm])
    assert torch.equal(run(planes(4.0 * x[:, s:])).float(), 4.0 * base.float())
