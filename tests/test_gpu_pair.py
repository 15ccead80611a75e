"""Fused ResBlock1 pair (ov_resblock_pair_f32, csrc/conv1d_pair.h) against
  * F.conv1d o leaky_relu o F.conv1d + x on the CPU (the reference's loop body, openvoice/modules.py:296-306), and
  * the two ov_conv1d_f32 launches it replaces -- bit for bit,
for every instantiated (C, K, dilation), ragged lengths, tile-boundary halos, utterance boundaries inside a
workgroup's run, runs that start mid-utterance (forced workgroup counts), the MRF sum / scale operands."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import PackedConv, launch_conv, launch_pair, pair_supported  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(c, k, d) for c in (32, 64) for k in (3, 7, 11) for d in (1, 3, 5)]


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _layers(c, k, d, seed=0):
    w1, b1 = _rand(c, c, k, seed=seed + 1, scale=(c * k) ** -0.5), _rand(c, seed=seed + 2, scale=0.1)
    w2, b2 = _rand(c, c, k, seed=seed + 3, scale=0.5 * (c * k) ** -0.5), _rand(c, seed=seed + 4, scale=0.1)
    return (w1, b1, w2, b2), PackedConv(w1, b1, DEV, K=k, dil=d), PackedConv(w2, b2, DEV, K=k, dil=1)


def _reference(x, w1, b1, w2, b2, k, d, add=None, scale=1.0):
    h = F.conv1d(F.leaky_relu(x, 0.1), w1, b1, dilation=d, padding=(k - 1) * d // 2)
    y = F.conv1d(F.leaky_relu(h, 0.1), w2, b2, padding=(k - 1) // 2) + x
    if add is not None:
        y = y + add
    return y * scale


def _two_launches(c1, c2, x, B, c, L, add=None, scale=1.0):
    t = torch.empty(B, c, L, device=DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(c1, x, 0, c * L, t, 0, c * L, B, L, in_slope=0.1)
    launch_conv(c2, t, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=x, res_bs=c * L, add=add, add_bs=c * L,
                scale=scale)
    return out


def _close(got, ref, tol=2e-5):
    got = got.cpu()
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) output elements"
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max-abs {err:.3e}"


@pytest.mark.parametrize("c,k,d", SHAPES)
def test_pair_matches_reference_and_two_launch_path(c, k, d):
    if not pair_supported(c, k, d):
        pytest.skip("no fused instance for this shape (the engine issues two launches)")
    B, L = 2, 1004                      # not a multiple of either step width (256 / 128)
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    x = _rand(B, c, L, seed=9)
    xd = x.to(DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_pair(c1, c2, xd, c * L, out, c * L, B, L)
    _close(out, _reference(x, w1, b1, w2, b2, k, d))
    assert torch.equal(out, _two_launches(c1, c2, xd, B, c, L))


@pytest.mark.parametrize("L", [4, 12, 124, 128, 132, 252, 256, 260, 516, 2048])
@pytest.mark.parametrize("c,k,d", [(32, 11, 5), (32, 3, 1), (64, 3, 5), (64, 7, 3)])
def test_pair_lengths_around_the_step_width(c, k, d, L):
    """Utterances shorter than the receptive field, shorter than one step, exactly one / two steps, one vector more."""
    if not pair_supported(c, k, d):
        pytest.skip("no fused instance")
    B = 3
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=L)
    x = _rand(B, c, L, seed=L + 5)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_pair(c1, c2, x.to(DEV), c * L, out, c * L, B, L)
    _close(out, _reference(x, w1, b1, w2, b2, k, d))


@pytest.mark.parametrize("nwg", [1, 2, 3, 5, 7, 64, 100000])
@pytest.mark.parametrize("c,k,d", [(32, 7, 3), (64, 3, 1)])
def test_pair_runs_starting_mid_utterance_and_spanning_utterances(c, k, d, nwg):
    """Forced workgroup counts: runs that begin in the middle of an utterance (warm-up step), runs that cross from one
    utterance into the next (left context must be zero again), more workgroups than steps."""
    if not pair_supported(c, k, d):
        pytest.skip("no fused instance")
    B, L = 3, 1540
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=nwg)
    x = _rand(B, c, L, seed=77)
    xd = x.to(DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_pair(c1, c2, xd, c * L, out, c * L, B, L, nwg=nwg)
    _close(out, _reference(x, w1, b1, w2, b2, k, d))
    assert torch.equal(out, _two_launches(c1, c2, xd, B, c, L))


@pytest.mark.parametrize("c,k,d", [(32, 11, 1), (64, 3, 3)])
def test_pair_mrf_sum_and_scale_in_place_on_the_accumulator(c, k, d):
    """Last pair of ResBlocks 2 and 3: out = (pair(x) + acc) / 3 written over acc (openvoice/models.py:282-286)."""
    if not pair_supported(c, k, d):
        pytest.skip("no fused instance")
    B, L = 2, 3000
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    x, acc = _rand(B, c, L, seed=3), _rand(B, c, L, seed=4)
    accd = acc.to(DEV)
    launch_pair(c1, c2, x.to(DEV), c * L, accd, c * L, B, L, add=accd, add_bs=c * L, scale=1.0 / 3.0)
    _close(accd, _reference(x, w1, b1, w2, b2, k, d, add=acc, scale=1.0 / 3.0))


def test_pair_padded_rows_and_poisoned_padding():
    """Rows `ld` apart with NaN in the padding columns: never read as data, never written."""
    c, k, d, B, L, ld = 32, 7, 5, 2, 700, 704
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    x = _rand(B, c, L, seed=1)
    xp = torch.full((B, c, ld), float("nan"))
    xp[:, :, :L] = x
    out = torch.full((B, c, ld), float("nan"), device=DEV)
    launch_pair(c1, c2, xp.to(DEV), c * ld, out, c * ld, B, L, ld=ld)
    _close(out[:, :, :L], _reference(x, w1, b1, w2, b2, k, d))
    assert torch.isnan(out[:, :, L:]).all()


def test_pair_refuses_what_it_cannot_do():
    c, k, d, B, L = 32, 3, 1, 1, 64
    _, c1, c2 = _layers(c, k, d)
    x = torch.zeros(B, c, L, device=DEV)
    with pytest.raises(_lib.OvError):       # in place
        launch_pair(c1, c2, x, c * L, x, c * L, B, L)
    with pytest.raises(_lib.OvError):       # rows not 16-byte aligned
        launch_pair(c1, c2, torch.zeros(B, c, 66, device=DEV), c * 66, torch.zeros(B, c, 66, device=DEV), c * 66, B, 66)
    _, c1b, c2b = _layers(128, 3, 1)
    xb = torch.zeros(1, 128, 64, device=DEV)
    assert not pair_supported(128, 3, 1)
    with pytest.raises(_lib.OvError):       # no instance: the caller must issue the two launches itself
        launch_pair(c1b, c2b, xb, 128 * 64, torch.zeros_like(xb), 128 * 64, 1, 64)


@pytest.mark.parametrize("name", ["resblock1_c32_k3", "resblock1_c32_k11", "resblock1_c64_k7"])
def test_three_fused_pairs_match_the_reference_resblock1_module(golden_dir, name):
    """tests/golden/resblock1_*.pt = the UNMODIFIED reference ``ResBlock1.forward`` (openvoice/modules.py:221-309) on a
    seeded input with random weight-norm parameters (oracle/make_resblock_golden.py).  Three fused launches -- weight-norm
    folded by ``params.effective_weight`` -- must reproduce the module; so must the two-launch path; the bf16 kernels
    within the bf16 path's tolerance."""
    import os
    from openvoice_amd.params import effective_weight
    rec = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    case, sd, x = rec["case"], rec["state_dict"], rec["x"]
    c, k, B, L = case["C"], case["K"], case["B"], case["L"]
    layers = [(PackedConv(effective_weight(sd, f"convs1.{i}"), sd[f"convs1.{i}.bias"], DEV, K=k, dil=d),
               PackedConv(effective_weight(sd, f"convs2.{i}"), sd[f"convs2.{i}.bias"], DEV, K=k, dil=1))
              for i, d in enumerate((1, 3, 5))]
    fused = all(pair_supported(c, k, d) for d in (1, 3, 5))
    cur = x.to(DEV)
    for n, (c1, c2) in enumerate(layers):
        nxt = torch.full_like(cur, float("nan"))
        if fused:
            launch_pair(c1, c2, cur, c * L, nxt, c * L, B, L)
        else:
            nxt = _two_launches(c1, c2, cur, B, c, L)
        if n == 0:
            _close(nxt, rec["after_pair0"])
        cur = nxt
    _close(cur, rec["out"], tol=3e-5)
    # bf16 twin (channels-last): within the bf16 path's stated tolerance of the fp32 module
    from openvoice_amd.bf16 import PackedConvBf16, launch_conv_bf16, launch_pair_bf16, pair_bf16_supported
    l16 = [(PackedConvBf16(effective_weight(sd, f"convs1.{i}"), sd[f"convs1.{i}.bias"], DEV, dil=d),
            PackedConvBf16(effective_weight(sd, f"convs2.{i}"), sd[f"convs2.{i}.bias"], DEV, dil=1))
           for i, d in enumerate((1, 3, 5))]
    cur = x.transpose(1, 2).contiguous().to(DEV, torch.bfloat16)
    for d, (c1, c2) in zip((1, 3, 5), l16):
        nxt = torch.full_like(cur, float("nan"))
        if pair_bf16_supported(c, k, d):
            launch_pair_bf16(c1, c2, cur, nxt)
        else:
            t = torch.empty_like(cur)
            launch_conv_bf16(c1, cur, t, in_slope=0.1)
            launch_conv_bf16(c2, t, nxt, in_slope=0.1, res=cur)
        cur = nxt
    err = (cur.float().cpu().transpose(1, 2) - rec["out"]).abs().max().item()
    assert err <= 3e-2 * rec["out"].abs().max().item(), err
