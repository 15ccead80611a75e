"""NOT COLLECTED by pytest (file name): checks for the LDS-DMA staging variant of the MRF convs
(openvoice_amd/csrc/conv1d_inst_g.hip, selected with loaders = OV_LOADERS_LDS_DMA = -1), written after the last full GPU
session of round 1.  The kernels themselves HAVE run on an MI355X through tools/micro/dma_check.hip (bit-identical to
the register-staged variant on six ragged shapes, 1-7 points slower: profiles/r01_s40_lds_dma_variant_check.txt);
these pytest checks have not.  With GPU time:

    python -m pytest tests/pending_gpu_lds_dma.py -q -m gpu -x          # (explicit path: pytest then collects it)
    python tools/bench_convs.py --loaders 0 -1 --modes plain1 res+add   # A/B against the shipped loaders

When green, rename to tests/test_gpu_lds_dma.py; when the A/B says so, make it the dispatcher's default."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402

DEV = "cuda:0"
LDS_DMA = -1


def _rand(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=gen)


@pytest.mark.parametrize("c", [32, 64, 128, 256])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)])
def test_lds_dma_staging_matches_reference_and_register_staging(c, k, d):
    """Same conv as test_resblock_conv_shapes, staged by global_load_lds: vs fp32 PyTorch within the usual bound,
    and bit-identical to the register-staged kernel of the same tile (same MFMA order; max(v, 0.1 v) == lrelu(v))."""
    B, L = 2, 1096 if c >= 128 else 2312          # multiples of 4, not of any tile
    x, res, add = _rand(B, c, L, seed=1), _rand(B, c, L, seed=2), _rand(B, c, L, seed=5)
    w, bias = _rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1)
    ref = (F.conv1d(F.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2) + res + add) / 3.0
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    kw = dict(in_slope=0.1, res=res.to(DEV), res_bs=c * L, add=add.to(DEV), add_bs=c * L, scale=1.0 / 3.0)
    xd = x.to(DEV)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, xd, 0, c * L, out, 0, c * L, B, L, loaders=LDS_DMA, **kw)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"C={c} k={k} d={d}: {err:.3e}"
    tile = 1 if c >= 128 else (2 if c == 64 else 4)          # the tile the DMA dispatcher picks: 128x128 / 64x256 / 32x256
    chunk = 32 if c >= 128 else 16
    same = torch.empty_like(out)
    launch_conv(layer, xd, 0, c * L, same, 0, c * L, B, L, tile=tile, chunk=chunk, **kw)
    assert torch.equal(out, same), f"C={c} k={k} d={d}: differs from the register-staged kernel"


@pytest.mark.parametrize("tpw", [0, 1, 3])
def test_lds_dma_staging_persistent_and_plain_grids(tpw):
    """Large enough for the persistent path (no residual) and several tiles per workgroup; no activation (slope 1)."""
    c, k, d, B, L = 128, 3, 1, 4, 128 * 70
    x = _rand(B, c, L, seed=1)
    w, bias = _rand(c, c, k, seed=3, scale=(c * k) ** -0.5), _rand(c, seed=4, scale=0.1)
    ref = F.conv1d(x, w, bias, dilation=d, padding=(k - 1) * d // 2)
    layer = PackedConv(w, bias, DEV, K=k, dil=d)
    out = torch.full((B, c, L), float("nan"), device=DEV)
    launch_conv(layer, x.to(DEV), 0, c * L, out, 0, c * L, B, L, in_slope=1.0, loaders=LDS_DMA, tiles_per_wg=tpw)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"tpw={tpw}: {err:.3e}"


def test_lds_dma_staging_refuses_what_it_cannot_do():
    c, k, B = 64, 3, 1
    layer = PackedConv(_rand(c, c, k, seed=1), None, DEV, K=k)
    for L, slope in ((2310, 0.1), (2312, 1.5), (2312, 0.0)):       # ragged rows; slopes where max != leaky ReLU
        x, out = torch.zeros(B, c, L, device=DEV), torch.zeros(B, c, L, device=DEV)
        with pytest.raises(_lib.OvError, match="OV_E_UNSUPPORTED"):
            launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, in_slope=slope, loaders=LDS_DMA)


def test_generator_with_lds_dma_equals_default(synth_sd):
    """The whole conversion with the MRF convs on the DMA variant: bit-identical waveform."""
    from openvoice_amd.engine import ConverterEngine
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    eng = ConverterEngine(synth_sd, CFG, 513, DEV, zero_g=False)
    gen = torch.Generator().manual_seed(3)
    B, T = 2, 40
    spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(DEV)
    lengths = torch.tensor([T, T - 5], device=DEV)
    g = [(0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV) for _ in range(2)]
    noise = torch.randn(B, 192, T, generator=gen).to(DEV)
    a = eng.voice_conversion(spec, lengths, g[0], g[1], tau=0.3, noise=noise)[0].clone()
    eng.mrf_loaders = LDS_DMA
    b = eng.voice_conversion(spec, lengths, g[0], g[1], tau=0.3, noise=noise)[0]
    assert torch.equal(a, b)
