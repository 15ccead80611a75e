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


@pytest.mark.parametrize("cin,cout,k,d,L,with_res", [(64, 192, 7, 1, 333, True), (64, 192, 3, 5, 50, False),
                                                     (96, 160, 11, 3, 401, True), (128, 320, 3, 1, 129, True)])
def test_conv_bf16_rectangular_partial_last_n_block(cin, cout, k, d, L, with_res):
    """Cout not a multiple of the 128 columns a workgroup owns: the last N-block has 1-2 live 32-column tiles (the dead
    ones read the packer's zero weight record, nothing of theirs is staged or stored), identity rounds of a partial
    block in 64- and 32-channel staging rounds (Cin % 64 == 0 or not), L shorter than a time tile."""
    B = 2
    x = _r(_rand(B, L, cin, seed=1))
    res, add = _r(_rand(B, L, cout, seed=2)), _r(_rand(B, L, cout, seed=3))
    w, bias = _r(_rand(cout, cin, k, seed=4, scale=(cin * k) ** -0.5)), _rand(cout, seed=5, scale=0.1)
    ref = F.conv1d(_r(F.leaky_relu(x, 0.1)).transpose(1, 2), w, bias, dilation=d, padding=(k - 1) * d // 2).transpose(1, 2)
    kw = {}
    if with_res:
        ref = ref + res + add
        kw = dict(res=res.to(DEV, torch.bfloat16), add=add.to(DEV, torch.bfloat16))
    layer = PackedConvBf16(w, bias, DEV, dil=d)
    out = torch.full((B, L, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    launch_conv_bf16(layer, x.to(DEV, torch.bfloat16), out, in_slope=0.1, **kw)
    err = (out.float().cpu() - ref).abs().max().item()
    bound = 1e-2 * max(1.0, ref.abs().max().item())
    assert err <= bound, f"{cin}->{cout} k={k} d={d} L={L}: {err:.3e} > {bound:.3e}"


def test_plain_conv_bf16_no_bias_no_residual():
    B, L, c, k = 1, 300, 64, 7
    x = _r(_rand(B, L, c, seed=1))
    w = _r(_rand(c, c, k, seed=2, scale=(c * k) ** -0.5))
    ref = F.conv1d(x.transpose(1, 2), w, None, padding=3).transpose(1, 2)
    layer = PackedConvBf16(w, None, DEV)
    out = torch.full((B, L, c), float("nan"), dtype=torch.bfloat16, device=DEV)
    launch_conv_bf16(layer, x.to(DEV, torch.bfloat16), out)
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("c,k,d", [(128, 7, 3), (64, 3, 1), (256, 11, 5)])
def test_conv_bf16_output_activation(c, k, d):
    """``out_slope``: the leaky ReLU of a tensor's only consumer applied before the output rounding (conv1 of a ResBlock
    pair on the two-launch path), and that consumer staging it with ``in_slope = 1`` (a plain copy)."""
    B, L = 2, 700
    x = _r(_rand(B, L, c, seed=1))
    w1, b1 = _r(_rand(c, c, k, seed=2, scale=(c * k) ** -0.5)), _rand(c, seed=3, scale=0.1)
    w2, b2 = _r(_rand(c, c, k, seed=4, scale=(c * k) ** -0.5)), _rand(c, seed=5, scale=0.1)
    t_ref = _r(F.leaky_relu(F.conv1d(_r(F.leaky_relu(x, 0.1)).transpose(1, 2), w1, b1, dilation=d,
                                     padding=(k - 1) * d // 2), 0.1))
    ref = (F.conv1d(t_ref, w2, b2, padding=(k - 1) // 2) + x.transpose(1, 2)).transpose(1, 2)
    c1, c2 = PackedConvBf16(w1, b1, DEV, dil=d), PackedConvBf16(w2, b2, DEV, dil=1)
    xd = x.to(DEV, torch.bfloat16)
    t = torch.full_like(xd, float("nan"))
    out = torch.full_like(xd, float("nan"))
    launch_conv_bf16(c1, xd, t, in_slope=0.1, out_slope=0.1)
    launch_conv_bf16(c2, t, out, in_slope=1.0, res=xd)
    assert (t.float().cpu() - t_ref.transpose(1, 2)).abs().max().item() <= 1e-2 * max(1.0, t_ref.abs().max().item())
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


def test_conv_transpose_and_batch_bias_bf16():
    """ups as a phase conv with (phase, channel) column order, and conv_pre's per-utterance bias over 512 columns
    (two N-blocks)."""
    from openvoice_amd.bf16 import _launch
    from openvoice_amd.engine import conv_transpose_as_conv
    B, L, cin, co, s = 2, 77, 64, 32, 8
    x = _r(_rand(B, L, cin, seed=1))
    w, b = _r(_rand(cin, co, 2 * s, seed=2, scale=(2 * cin) ** -0.5)), _rand(co, seed=3, scale=0.1)
    ref = F.conv_transpose1d(_r(F.leaky_relu(x, 0.1)).transpose(1, 2), w, b, stride=s, padding=s // 2).transpose(1, 2)
    wc = conv_transpose_as_conv(w, s).reshape(co, s, cin, 3).transpose(0, 1).reshape(s * co, cin, 3)
    layer = PackedConvBf16(wc, b.repeat(s), DEV)
    out = torch.full((B, L * s, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    _launch(layer, x.to(DEV, torch.bfloat16), out, L, in_slope=0.1, phase_s=s)
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
    # 512 output columns with a per-utterance bias
    cin, cout, k = 192, 512, 7
    x = _r(_rand(B, L, cin, seed=4))
    w, bb = _r(_rand(cout, cin, k, seed=5, scale=(cin * k) ** -0.5)), _rand(B, cout, seed=6)
    ref = F.conv1d(x.transpose(1, 2), w, None, padding=3).transpose(1, 2) + bb[:, None, :]
    layer = PackedConvBf16(w, None, DEV)
    out = torch.full((B, L, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    _launch(layer, x.to(DEV, torch.bfloat16), out, L, bias=bb.to(DEV), bias_bstride=cout)
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,T,per_item", [(1, 33, False), (3, 70, True)])
def test_generator_bf16_against_fp32_oracle(synth_sd, B, T, per_item):
    """The whole generator with bf16 activations against the fp32 oracle (reference: openvoice/models.py:272-291).
    bf16 keeps 8 significant bits per stored activation; over the ~80 layers of the generator the waveform
    (|o| <= 1) lands within a few 1e-2 of the fp32 result.  Measured: max-abs 0.007-0.011, relative RMS 0.6 %;
    stated tolerance of this path: max-abs 3e-2, relative RMS error 1.5 %."""
    from openvoice_amd.bf16 import GeneratorBf16
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    from oracle import vc_oracle
    gen = torch.Generator().manual_seed(T)
    z = torch.randn(B, 192, T, generator=gen)
    g = 0.3 * torch.randn(B if per_item else 1, 256, 1, generator=gen)
    with torch.no_grad():
        ref = vc_oracle.generator(synth_sd, z, g, CFG)
    dec = GeneratorBf16(synth_sd, CFG, DEV)
    o = dec.decode(z.to(DEV), g.to(DEV))
    torch.cuda.synchronize()
    assert o.shape == ref.shape and o.dtype == torch.float32
    err = (o.cpu() - ref).abs()
    rel_rms = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"bf16 generator B={B} T={T}: max-abs {err.max().item():.4f}, rel RMS {rel_rms:.4f}, |ref|max {ref.abs().max().item():.3f}")
    assert err.max().item() <= 3e-2 and rel_rms <= 1.5e-2


def test_voice_conversion_with_the_opt_in_bf16_generator(synth_sd):
    """engine.use_bf16_generator(): enc_q and flow stay fp32 (latents unchanged to fp32 round-off), only the generator
    runs in bf16; the waveform stays within the bf16 path's stated tolerance of the fp32 oracle."""
    from openvoice_amd.models import SynthesizerTrn
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    from oracle import vc_oracle
    gen = torch.Generator().manual_seed(11)
    B, T = 2, 50
    spec = torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]
    g1, g2 = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
    noise = torch.randn(B, 192, T, generator=gen)
    lengths = torch.tensor([T, 31])
    with torch.no_grad():
        o_r, _, (_, _, zh_r) = vc_oracle.voice_conversion(synth_sd, CFG, spec, lengths, g1, g2, 0.3, noise)
    m = SynthesizerTrn(0, 513, n_speakers=0, **CFG)
    m.load_state_dict(synth_sd, strict=True)
    m = m.to(DEV).eval()
    m.engine().use_bf16_generator(True)
    o, _, (_, _, z_hat) = m.voice_conversion(spec.to(DEV), lengths.to(DEV), g1.to(DEV), g2.to(DEV), tau=0.3, noise=noise.to(DEV))
    assert (z_hat.cpu() - zh_r).abs().max().item() <= 2e-4
    err = (o.cpu() - o_r).abs().max().item()
    assert 1e-5 < err <= 3e-2, err          # genuinely the bf16 generator, and within its tolerance
    m.engine().use_bf16_generator(False)
    o32 = m.voice_conversion(spec.to(DEV), lengths.to(DEV), g1.to(DEV), g2.to(DEV), tau=0.3, noise=noise.to(DEV))[0]
    assert (o32.cpu() - o_r).abs().max().item() <= 1e-3


def test_concurrent_resblock_chains_are_bit_identical_to_the_serial_order(synth_sd):
    """``GeneratorBf16.chain_streams``: the three ResBlock chains of a stage on three HIP streams (the k = 3 chain is
    HBM-bound, the k = 11 chain matrix-bound: side by side they fill each other's idle resource); the MRF sum keeps its
    order through events, so the waveform equals the one-stream order bit for bit -- fused and unfused pairs, twice in
    a row (buffers are reused across calls), and under a non-default current stream."""
    from openvoice_amd.bf16 import GeneratorBf16
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(3, 192, 90, generator=gen).to(DEV)
    g = (0.3 * torch.randn(3, 256, 1, generator=gen)).to(DEV)
    dec = GeneratorBf16(synth_sd, CFG, DEV)
    for fuse in (True, False):
        dec.fuse_pairs = fuse
        dec.chain_streams = 1
        serial = dec.decode(z, g).clone()
        dec.chain_streams = 3
        a = dec.decode(z, g).clone()
        b = dec.decode(z, g).clone()
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            c = dec.decode(z, g).clone()
        side.synchronize()
        torch.cuda.synchronize()
        assert torch.isfinite(serial).all()
        assert torch.equal(a, serial) and torch.equal(b, serial) and torch.equal(c, serial), f"fuse_pairs={fuse}"
