"""The torch binding of the C ABI (openvoice_amd/csrc/torch_shim.cpp, `torch.ops.openvoice_amd.*`):
same kernels, same arguments as the ctypes binding -- results must be bit-identical -- with the current-stream pickup and
TORCH_CHECK error behaviour of a torch extension."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402
from openvoice_amd.models import SynthesizerTrn  # noqa: E402
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG  # noqa: E402

DEV = "cuda:0"


def test_conv_through_torch_ops_equals_ctypes(monkeypatch):
    c, k, d, B, L = 128, 7, 3, 2, 1000
    gen = torch.Generator().manual_seed(0)
    layer = PackedConv(torch.randn(c, c, k, generator=gen) * (c * k) ** -0.5, torch.randn(c, generator=gen) * 0.1, DEV,
                       K=k, dil=d)
    x = torch.randn(B, c, L, generator=gen).to(DEV)
    res = torch.randn(B, c, L, generator=gen).to(DEV)
    outs = []
    for binding in ("ctypes", "torch"):
        monkeypatch.setenv("OPENVOICE_AMD_BINDING", binding)
        out = torch.full((B, c, L), float("nan"), device=DEV)
        launch_conv(layer, x, 0, c * L, out, 0, c * L, B, L, in_slope=0.1, res=res, res_bs=c * L, scale=0.5)
        outs.append(out)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_launch_follows_the_current_torch_stream(monkeypatch):
    """The shim launches on c10::hip::getCurrentHIPStream(): work issued under torch.cuda.stream(s) is ordered with s."""
    monkeypatch.setenv("OPENVOICE_AMD_BINDING", "torch")
    ops = _lib.torch_ops()
    side = torch.cuda.Stream(DEV)
    x = torch.randn(64, 256, device=DEV)
    w, b = torch.randn(512, 256, device=DEV), torch.randn(512, device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        big = torch.randn(1 << 24, device=DEV).sin().sum()      # keeps `side` busy in front of the op
        y = ops.linear(x, w, b)
        side_evt = side.record_event()
    side_evt.synchronize()
    assert torch.allclose(y, x @ w.t() + b, atol=1e-3) and bool(torch.isfinite(big))


def test_errors_are_runtime_errors(monkeypatch):
    ops = _lib.torch_ops()
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3))
    with pytest.raises(RuntimeError, match="float32"):
        ops.linear(torch.zeros(2, 4, device=DEV, dtype=torch.float64), torch.zeros(3, 4, device=DEV), torch.zeros(3, device=DEV))
    x = torch.zeros(1, 32, 66, device=DEV)        # rows not 16-byte aligned: OV_E_ALIGN from the library
    w = torch.zeros(8192, device=DEV)
    with pytest.raises(RuntimeError, match="ov_resblock_pair_f32 failed"):
        ops.resblock_pair(x, w, w, w, w, torch.zeros_like(x), None, 1, 32, 66, 66, 3, 1, 32 * 66, 32 * 66, 0, 0.1, 1.0)


def test_whole_conversion_through_the_torch_binding(synth_sd, monkeypatch):
    """Every launch of a conversion (mask, conditioning GEMVs, all convs, fused pairs, conv_post) through torch.ops."""
    model = SynthesizerTrn(0, 513, n_speakers=0, zero_g=True, **CONVERTER_MODEL_CONFIG)
    model.load_state_dict(synth_sd, strict=True)
    model = model.to(DEV).eval()
    gen = torch.Generator().manual_seed(3)
    B, T = 3, 70
    spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(DEV)
    lengths = torch.tensor([T, 50, 9], device=DEV)
    g1, g2 = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV), (0.3 * torch.randn(B, 256, 1, generator=gen)).to(DEV)
    noise = torch.randn(B, 192, T, generator=gen).to(DEV)
    outs = []
    for binding in ("ctypes", "torch"):
        monkeypatch.setenv("OPENVOICE_AMD_BINDING", binding)
        o, m, lat = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise)
        torch.cuda.synchronize()
        outs.append((o.clone(), m.clone(), [t.clone() for t in lat]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))
