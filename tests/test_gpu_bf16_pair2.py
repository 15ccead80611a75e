"""Second-generation fused bf16 ResBlock1 pair (ov_resblock_pair2_bf16cl, csrc/conv1d_bf16_pair2.hip: C = 32 / 64 / 128,
activations stored ACTIVATED in HBM) against fp32 PyTorch on the same bf16-rounded operands with every rounding of the
kernel mirrored (bound: 1e-2 of scale, i.e. output rounding): every (C, K, dilation), lengths around the step height,
utterance boundaries inside a run, runs starting mid-utterance (forced workgroup counts), the MRF operands, the output
activation.  reference: openvoice/modules.py:296-306, models.py:280-286."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd.bf16 import PackedConvBf16, launch_pair2_bf16, pair2_bf16_supported  # noqa: E402

DEV = "cuda:0"
SLOPE = 0.1


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _r(t):
    return t.to(torch.bfloat16).float()


def _layers(c, k, d, seed=0):
    w1, b1 = _r(_rand(c, c, k, seed=seed + 1, scale=(c * k) ** -0.5)), _rand(c, seed=seed + 2, scale=0.1)
    w2, b2 = _r(_rand(c, c, k, seed=seed + 3, scale=0.5 * (c * k) ** -0.5)), _rand(c, seed=seed + 4, scale=0.1)
    return (w1, b1, w2, b2), PackedConvBf16(w1, b1, DEV, dil=d), PackedConvBf16(w2, b2, DEV, dil=1)


def _reference(xa, w1, b1, w2, b2, k, d, add=None, scale=1.0, out_slope=1.0):
    """xa (B, L, C): the ACTIVATED input, already bf16-rounded; mirrors every rounding of the kernel."""
    xat = xa.transpose(1, 2)
    t = _r(F.leaky_relu(F.conv1d(xat, w1, b1, dilation=d, padding=(k - 1) * d // 2), SLOPE))
    inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(SLOPE, dtype=torch.float32)
    x_raw = torch.where(xat >= 0, xat, xat * inv)
    y = F.conv1d(t, w2, b2, padding=(k - 1) // 2) + x_raw
    if add is not None:                      # the running sum is added to the ROUNDED pair output, on the way out
        y = F.leaky_relu((_r(y) + add.transpose(1, 2)) * scale, out_slope)
    else:
        y = F.leaky_relu(y * scale, out_slope)
    return y.transpose(1, 2)


def _check(out, ref):
    assert torch.isfinite(out.float()).all(), "unwritten (NaN-poisoned) output elements"
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 1e-2 * max(1.0, ref.abs().max().item()), err


def _act_input(B, L, c, seed):
    return _r(F.leaky_relu(_rand(B, L, c, seed=seed), SLOPE))


@pytest.mark.parametrize("c", [32, 64, 128])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)])
def test_pair2_matches_reference(c, k, d):
    assert pair2_bf16_supported(c, k, d)
    B, L = 2, 1531
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d)
    xa = _act_input(B, L, c, 9)
    xd = xa.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out)
    _check(out, _reference(xa, w1, b1, w2, b2, k, d))


@pytest.mark.parametrize("L", [1, 5, 127, 128, 129, 255, 256, 257, 383, 384, 511, 512, 513, 600, 1030])
@pytest.mark.parametrize("c,k,d", [(128, 11, 5), (128, 3, 1), (64, 7, 3), (64, 11, 5), (32, 3, 1), (32, 11, 5)])
def test_pair2_lengths_around_the_step_height(c, k, d, L):
    B = 3
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=L)
    xa = _act_input(B, L, c, L + 5)
    xd = xa.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out)
    _check(out, _reference(xa, w1, b1, w2, b2, k, d))


@pytest.mark.parametrize("nwg", [1, 2, 3, 5, 7, 16])
@pytest.mark.parametrize("c,k,d", [(128, 7, 5), (64, 11, 3), (32, 7, 1)])
def test_pair2_runs_that_start_mid_utterance_and_span_utterances(c, k, d, nwg):
    """Forced workgroup counts cut the (utterance, step) list at arbitrary places: runs that start mid-utterance take
    the warm-up pseudo-step, runs that cross an utterance boundary restart the t context from zeros."""
    B, L = 3, 1100 if c > 32 else 2300
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=nwg)
    xa = _act_input(B, L, c, 40 + nwg)
    xd = xa.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out, nwg=nwg)
    _check(out, _reference(xa, w1, b1, w2, b2, k, d))
    ref = out.clone()
    out.fill_(float("nan"))
    launch_pair2_bf16(c1, c2, xd, out)                 # the default workgroup count: same rows, same arithmetic
    assert torch.equal(out, ref)


@pytest.mark.parametrize("c,k,d", [(128, 3, 3), (128, 11, 1), (64, 3, 5), (64, 7, 1), (32, 11, 5), (32, 3, 1)])
def test_pair2_mrf_operands_and_output_activation(c, k, d):
    B, L = 2, 777 if c > 32 else 1500
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=3)
    xa = _act_input(B, L, c, 21)
    add = _r(_rand(B, L, c, seed=22))
    xd, addd = xa.to(DEV, torch.bfloat16), add.to(DEV, torch.bfloat16)
    out = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out, add=addd, scale=1.0 / 3.0)
    _check(out, _reference(xa, w1, b1, w2, b2, k, d, add=add, scale=1.0 / 3.0))
    # `add` may alias `out` (the running sum is updated in place by the last pair of a chain)
    acc = addd.clone()
    launch_pair2_bf16(c1, c2, xd, acc, add=acc, scale=1.0 / 3.0)
    assert torch.equal(acc, out)
    # an intermediate pair stores its output activated for the next one
    out2 = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out2, out_slope=SLOPE)
    _check(out2, _reference(xa, w1, b1, w2, b2, k, d, out_slope=SLOPE))
    # the MRF mean is stored activated for the next stage's ConvTranspose (openvoice/models.py:278-279)
    out3 = torch.full_like(xd, float("nan"))
    launch_pair2_bf16(c1, c2, xd, out3, add=addd, scale=1.0 / 3.0, out_slope=SLOPE)
    _check(out3, _reference(xa, w1, b1, w2, b2, k, d, add=add, scale=1.0 / 3.0, out_slope=SLOPE))


def test_pair2_chain_of_three_pairs_matches_raw_residual_chain():
    """Three pairs back to back on activated tensors = ResBlock1.forward (modules.py:296-306) on the raw tensor, within
    bf16 storage rounding of the intermediates."""
    c, k, B, L = 128, 7, 2, 900
    layers = [_layers(c, k, d, seed=10 * d) for d in (1, 3, 5)]
    x = _r(_rand(B, L, c, seed=77))
    cur = F.leaky_relu(x, SLOPE).to(DEV, torch.bfloat16)
    bufs = [torch.empty_like(cur), torch.empty_like(cur)]
    ref = x.transpose(1, 2)
    for n, ((w1, b1, w2, b2), c1, c2) in enumerate(layers):
        d = (1, 3, 5)[n]
        last = n == 2
        launch_pair2_bf16(c1, c2, cur, bufs[n & 1], out_slope=1.0 if last else SLOPE)
        cur = bufs[n & 1]
        t = F.conv1d(F.leaky_relu(ref, SLOPE), w1, b1, dilation=d, padding=(k - 1) * d // 2)
        ref = F.conv1d(F.leaky_relu(t, SLOPE), w2, b2, padding=(k - 1) // 2) + ref
    err = (cur.float().cpu() - ref.transpose(1, 2)).abs().max().item()
    assert err <= 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("c,k,d", [(128, 11, 5), (128, 3, 1), (64, 7, 3), (32, 3, 5)])
def test_pair2_result_does_not_depend_on_how_steps_are_dealt_to_workgroups(c, k, d):
    """One persistent workgroup per CU walks a run of steps; a run that starts mid-utterance recomputes the t context
    of the tile before it (warm pseudo-step).  Whatever the cut -- one workgroup for everything, 5, 37, one per CU --
    the output bits are the same, in every epilogue mode (activated, running sum + scale, plain, scaled)."""
    B, L = 3, 1900
    (w1, b1, w2, b2), c1, c2 = _layers(c, k, d, seed=5)
    xa = _act_input(B, L, c, 61)
    add = _r(_rand(B, L, c, seed=62)).to(DEV, torch.bfloat16)
    xd = xa.to(DEV, torch.bfloat16)
    for kw in (dict(out_slope=SLOPE), dict(add=add, scale=1.0 / 3.0), dict(), dict(scale=1.0 / 3.0)):
        a = torch.full_like(xd, float("nan"))
        launch_pair2_bf16(c1, c2, xd, a, **kw)
        assert torch.isfinite(a.float()).all()
        for nwg in (1, 5, 37):
            b = torch.full_like(xd, float("nan"))
            launch_pair2_bf16(c1, c2, xd, b, nwg=nwg, **kw)
            assert torch.equal(a, b), (kw.keys(), nwg)


def test_pair2_rejects_bad_arguments():
    from openvoice_amd import _lib
    (w1, b1, w2, b2), c1, c2 = _layers(64, 3, 1)
    x = torch.zeros(1, 64, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.OvError):
        launch_pair2_bf16(c1, c2, x, x)                               # out aliases x
    with pytest.raises(_lib.OvError):
        launch_pair2_bf16(c1, c2, x, torch.empty_like(x), slope=0.0)  # the residual inverse needs slope > 0
    assert not pair2_bf16_supported(16, 3, 1) and not pair2_bf16_supported(256, 3, 1)
