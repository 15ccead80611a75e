"""Fused bf16 ResBlock1 pair (ov_resblock_pair_bf16cl, csrc/conv1d_bf16_pair.hip) against
  * the two ov_conv1d_bf16cl launches it replaces -- bit for bit (the intermediate is rounded to bf16 at the same point), and
  * fp32 PyTorch on the same bf16-rounded operands, intermediate rounded to bf16 like the kernels do (bound 1e-2 of scale),
for every (C, K, dilation), ragged lengths, utterance boundaries inside a run, runs starting mid-utterance, the MRF
operands.  reference: openvoice/modules.py:296-306."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.bf16 import PackedConvBf16, launch_conv_bf16, launch_pair_bf16, pair_bf16_supported  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _r(t):
    return t.to(torch.bfloat16).float()


def _layers(c, k, d, seed=0):
    w1, b1 = _r(_rand(c, c, k, seed=seed + 1, scale=(c * k) ** -0.5)), _rand(c, seed=seed + 2, scale=0.1)
    w2, b2 = _r(_rand(c, c, k, seed=seed + 3, scale=0.5 * (c * k) ** -0.5)), _rand(c, seed=seed + 4, scale=0.1)
    return (w1, b1, w2, b2), PackedConvBf16(w1, b1, DEV, dil=d), PackedConvBf16(w2, b2, DEV, dil=1)


def _reference(x, w1, b1, w2, b2, k, d, add=None, scale=1.0):
    """x (B, L, C) already bf16-rounded; mirrors every rounding of the kernels."""
    xt = x.transpose(1, 2)
    t = _r(F.conv1d(_r(F.leaky_relu(xt, 0.1)), w1, b1, dilation=d, padding=(k - 1) * d // 2))
    y = F.conv1d(_r(F.leaky_relu(t, 0.1)), w2, b2, padding=(k - 1) // 2) + xt
    if add is not None:
        y = y + add.transpose(1, 2)
    return (y * scale).transpose(1, 2)


def _two_launches(c1, c2, x, add=None, scale=1.0):
    t = torch.empty_like(x)
    out = torch.full_like(x, float("nan"))
    launch_conv_bf16(c1, x, t, in_slope=0.1)
    launch_conv_bf16(c2, t, out, in_slope=0.1, res=x, add=add, scale=scale)
    return out


def _check(out, ref):
    assert torch.isfinite(out.float()).all(), "unwritten (NaN-poisoned) output elements"
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 1e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("c", [32, 64])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)])
def test_pair_bf16_matches_two_launch_path_and_reference(c, k, d):
    if not pair_bf16_supported(c, k, d):
        pytest.skip("no fused instance: both convs' weights must fit in LDS beside the tiles (C = 64: K = 3 only)")
    B, L = 2, 1531
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    x = _r(_rand(B, L, c, seed=9))
    xd = x.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair_bf16(c1, c2, xd, out)
    _check(out, _reference(x, w1, b1, w2, b2, k, d))
    assert torch.equal(out, _two_launches(c1, c2, xd))


@pytest.mark.parametrize("L", [1, 5, 127, 128, 129, 255, 256, 257, 600])
@pytest.mark.parametrize("c,k,d", [(32, 11, 5), (32, 3, 1), (64, 3, 3)])
def test_pair_bf16_lengths_around_the_step_height(c, k, d, L):
    B = 3
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=L)
    x = _r(_rand(B, L, c, seed=L + 5))
    xd = x.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair_bf16(c1, c2, xd, out)
    _check(out, _reference(x, w1, b1, w2, b2, k, d))
    assert torch.equal(out, _two_launches(c1, c2, xd))


@pytest.mark.parametrize("nwg", [1, 2, 3, 5, 7, 100000])
@pytest.mark.parametrize("c,k,d", [(32, 7, 3), (64, 3, 1), (32, 11, 5)])
def test_pair_bf16_runs_starting_mid_utterance_and_spanning_utterances(c, k, d, nwg):
    B, L = 3, 1540
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=nwg)
    x = _r(_rand(B, L, c, seed=77))
    xd = x.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair_bf16(c1, c2, xd, out, nwg=nwg)
    assert torch.equal(out, _two_launches(c1, c2, xd))


@pytest.mark.parametrize("c,k,d", [(32, 11, 1), (64, 3, 3)])
def test_pair_bf16_mrf_sum_and_scale_in_place_on_the_accumulator(c, k, d):
    B, L = 2, 3000
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    x, acc = _r(_rand(B, L, c, seed=3)), _r(_rand(B, L, c, seed=4))
    xd, accd = x.to(DEV, torch.bfloat16), acc.to(DEV, torch.bfloat16)
    want = _two_launches(c1, c2, xd, add=accd, scale=1.0 / 3.0)
    launch_pair_bf16(c1, c2, xd, accd, add=accd, scale=1.0 / 3.0)
    _check(accd, _reference(x, w1, b1, w2, b2, k, d, add=acc, scale=1.0 / 3.0))
    assert torch.equal(accd, want)


def test_pair_bf16_refuses_what_it_cannot_do():
    _, c1, c2 = _layers(32, 3, 1)
    x = torch.zeros(1, 64, 32, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.OvError):
        launch_pair_bf16(c1, c2, x, x)
    assert not pair_bf16_supported(128, 3, 1)


@pytest.mark.parametrize("c,k,d", [(32, 3, 1), (32, 11, 5), (64, 3, 3)])
def test_activated_store_two_launch_path_agrees_with_the_fused_pair_within_bf16_rounding(c, k, d):
    """``GeneratorBf16`` runs the pairs WITHOUT a fused instance as two launches that store the intermediate ACTIVATED
    (``out_slope``: t = bf16(lrelu(v)), one rounding) while the fused kernel -- and the plain two-launch sequence above it
    is bit-identical to -- rounds, activates in fp32 and rounds again (t = bf16(lrelu(bf16(v)))).  The two forms are
    NOT bit-identical (negative values can differ by one bf16 ulp of t); both are within the bf16 bound of the fp32
    reference.  The fused form is canonical for the stages it covers; stages are never mixed within one pair."""
    B, L = 2, 2000
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=5)
    x = _r(_rand(B, L, c, seed=6))
    xd = x.to(DEV, torch.bfloat16)
    fused = torch.full_like(xd, float("nan"))
    launch_pair_bf16(c1, c2, xd, fused)
    t = torch.empty_like(xd)
    two = torch.full_like(xd, float("nan"))
    launch_conv_bf16(c1, xd, t, in_slope=0.1, out_slope=0.1)          # the engine's sequence (bf16.py decode)
    launch_conv_bf16(c2, t, two, in_slope=1.0, res=xd)
    ref = _reference(x, w1, b1, w2, b2, k, d)
    _check(two, ref)
    _check(fused, ref)
    scale = max(1.0, ref.abs().max().item())
    diff = (two.float() - fused.float()).abs().max().item()
    assert diff <= 2 ** -6 * scale, diff          # two bf16 ulps of the output scale
