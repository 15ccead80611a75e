"""The torch binding of the C ABI (openvoice_amd/csrc/torch_shim.cpp, `torch.ops.openvoice_amd.*`; SURVEY.md section 8b;
the package default): same kernels, same arguments as the ctypes binding -- results must be bit-identical -- with the
current-stream pickup and TORCH_CHECK error behaviour of a torch extension, and NO ctypes anywhere on the path: the
whole-path tests below make ``_lib.load`` (the only ctypes.CDLL in the package) raise."""
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
        y = torch.empty(64, 512, device=DEV)
        ops.linear_f32(x, w, b, y, 64, 512, 256)
        side_evt = side.record_event()
    side_evt.synchronize()
    assert torch.allclose(y, x @ w.t() + b, atol=1e-3) and bool(torch.isfinite(big))


def test_errors_are_runtime_errors(monkeypatch):
    ops = _lib.torch_ops()
    y = torch.zeros(2, 3, device=DEV)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.linear_f32(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3), torch.zeros(2, 3), 2, 3, 4)
    with pytest.raises(RuntimeError, match="float32"):
        ops.linear_f32(torch.zeros(2, 4, device=DEV, dtype=torch.float64), torch.zeros(3, 4, device=DEV),
                       torch.zeros(3, device=DEV), y, 2, 3, 4)
    with pytest.raises(RuntimeError, match="OV_E_BADARG"):        # a required pointer left out: the library's own check
        ops.linear_f32(None, torch.zeros(3, 4, device=DEV), torch.zeros(3, device=DEV), y, 2, 3, 4)
    x = torch.zeros(1, 32, 66, device=DEV)        # rows not 16-byte aligned: OV_E_ALIGN from the library
    w = torch.zeros(8192, device=DEV)
    with pytest.raises(RuntimeError, match="ov_resblock_pair_f32 failed"):
        ops.resblock_pair_f32(x, w, w, w, w, torch.zeros_like(x), None, None, None,
                              [1, 32, 66, 66, 3, 1, 0, 32 * 66, 32 * 66, 0, 1], [0.1, 1.0])
    # through the package's call layer both bindings raise the same exception type
    monkeypatch.setenv("OPENVOICE_AMD_BINDING", "torch")
    with pytest.raises(_lib.OvError, match="OV_E_BADARG"):
        _lib.call("ov_linear_f32", None, torch.zeros(3, 4, device=DEV), torch.zeros(3, device=DEV), y, 2, 3, 4)


def _no_ctypes(monkeypatch):
    """Select the torch binding and make the package's only ctypes entry raise: a launch that bypasses the shim fails."""
    monkeypatch.setenv("OPENVOICE_AMD_BINDING", "torch")

    def refuse():
        raise AssertionError("ctypes binding used while OPENVOICE_AMD_BINDING=torch")
    monkeypatch.setattr(_lib, "load", refuse)


def test_whole_conversion_through_the_torch_binding(synth_sd, monkeypatch):
    """Every launch of a conversion (spectrogram framing, mask, conditioning GEMVs, fused WaveNet layers, all convs,
    fused pairs, conv_post) and the load-time weight packers through torch.ops, bit-identical to ctypes."""
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
    monkeypatch.setenv("OPENVOICE_AMD_BINDING", "ctypes")
    o, m, lat = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise)
    torch.cuda.synchronize()
    outs.append((o.clone(), m.clone(), [t.clone() for t in lat]))
    _no_ctypes(monkeypatch)
    model._engine = None          # rebuild the engine: the weight packers run through the shim too
    o, m, lat = model.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise)
    torch.cuda.synchronize()
    outs.append((o.clone(), m.clone(), [t.clone() for t in lat]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))


def test_spectrogram_extract_se_and_bf16_generator_without_ctypes(synth_sd, monkeypatch):
    """``spectrogram_torch`` (ov_frame_hops_f32 + framing conv), ``ref_enc`` (layernorm_freq, conv2d_s2_relu x6, GRU,
    linear) and the opt-in bf16 generator with the ctypes binding unreachable; results equal the ctypes binding's."""
    from openvoice_amd.mel_processing import _native, spectrogram_torch
    gen = torch.Generator().manual_seed(5)
    wave = (0.3 * torch.randn(2, 22050, generator=gen)).to(DEV)
    z = torch.randn(2, 192, 40, generator=gen).to(DEV)
    g = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV)
    res = []
    for binding in ("ctypes", "torch"):
        if binding == "torch":
            _no_ctypes(monkeypatch)
        else:
            monkeypatch.setenv("OPENVOICE_AMD_BINDING", binding)
        _native.clear()
        model = SynthesizerTrn(0, 513, n_speakers=0, **CONVERTER_MODEL_CONFIG)
        model.load_state_dict(synth_sd, strict=True)
        model = model.to(DEV).eval()
        spec = spectrogram_torch(wave, 1024, 22050, 256, 1024, center=False)
        se = model.ref_enc(spec.transpose(1, 2))
        eng = model.engine().use_bf16_generator(True)
        o16 = eng.generator_bf16.decode(z, g)
        torch.cuda.synchronize()
        res.append((spec.clone(), se.clone(), o16.clone()))
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_tts_infer_without_ctypes(synth_tts_sd, monkeypatch):
    """``SynthesizerTrn.infer`` (embed, relative attention, channel LayerNorm, depthwise conv, spline, durations, prior
    expansion + the converter's flow and generator) with the ctypes binding unreachable."""
    gen = torch.Generator().manual_seed(9)
    tokens = torch.randint(0, 68, (2, 21), generator=gen)
    lengths = torch.tensor([21, 13])
    sid = torch.tensor([1, 4])
    noise_w = torch.randn(2, 2, 21, generator=gen)
    noise_z = torch.randn(2, 192, 400, generator=gen)
    res = []
    for binding in ("ctypes", "torch"):
        if binding == "torch":
            _no_ctypes(monkeypatch)
        else:
            monkeypatch.setenv("OPENVOICE_AMD_BINDING", binding)
        model = SynthesizerTrn(68, 513, n_speakers=10, **CONVERTER_MODEL_CONFIG)
        model.load_state_dict(synth_tts_sd, strict=True)
        model = model.to(DEV).eval()
        o, attn, y_mask, _ = model.infer(tokens.to(DEV), lengths.to(DEV), sid=sid.to(DEV), noise_scale=0.667,
                                         noise_scale_w=0.6, length_scale=1.0, noise_w=noise_w.to(DEV),
                                         noise_z=noise_z.to(DEV))
        torch.cuda.synchronize()
        res.append((o.clone(), attn.clone(), y_mask.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
