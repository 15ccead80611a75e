"""bf16 channels-last generator convs (BASELINE.json configs[4]) against fp32 PyTorch evaluated on the SAME
bf16-rounded operands: inputs and weights are rounded to bf16 first (and the leaky-ReLU output re-rounded, as the
kernel does while staging), the reference then accumulates in fp32; what remains is the output rounding to bf16
(half an ulp = 2^-9 relative) plus summation order.  Bound: 1e-2 of the output scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd.bf16 import PackedConvBf16, launch_conv_bf16  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _r(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("c", [32, 64, 128, 256])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 5), (7, 3), (11, 1), (11, 5)])
def test_resblock_conv_bf16(c, k, d):
    B, L = 2, 1000 if c >= 128 else 1531          # not a multiple of any tile
    x, res, add = _r(_rand(B, L, c, seed=1)), _r(_rand(B, L, c, seed=2)), _r(_rand(B, L, c, seed=3))
    w, bias = _r(_rand(c, c, k, seed=4, scale=(c * k) ** -0.5)), _rand(c, seed=5, scale=0.1)
    xin = _r(F.leaky_relu(x, 0.1))
    ref = (F.conv1d(xin.transpose(1, 2), w, bias, dilation=d, padding=(k - 1) * d // 2).transpose(1, 2)
           + res + add) / 3.0
    layer = PackedConvBf16(w, bias, DEV, dil=d)
    out = torch.full((B, L, c), float("nan"), dtype=torch.bfloat16, device=DEV)
    launch_conv_bf16(layer, x.to(DEV, torch.bfloat16), out, in_slope=0.1, scale=1.0 / 3.0,
                     res=res.to(DEV, torch.bfloat16), add=add.to(DEV, torch.bfloat16))
    err = (out.float().cpu() - ref).abs().max().item()
    bound = 1e-2 * max(1.0, ref.abs().max().item())
    assert err <= bound, f"C={c} k={k} d={d}: {err:.3e} > {bound:.3e}"


def test_plain_conv_bf16_no_bias_no_residual():
    B, L, c, k = 1, 300, 64, 7
    x = _r(_rand(B, L, c, seed=1))
    w = _r(_rand(c, c, k, seed=2, scale=(c * k) ** -0.5))
    ref = F.conv1d(x.transpose(1, 2), w, None, padding=3).transpose(1, 2)
    layer = PackedConvBf16(w, None, DEV)
    out = torch.full((B, L, c), float("nan"), dtype=torch.bfloat16, device=DEV)
    launch_conv_bf16(layer, x.to(DEV, torch.bfloat16), out)
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
