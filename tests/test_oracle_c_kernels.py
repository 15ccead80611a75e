"""The plain-C, loop-level part of the oracle (oracle/vc_kernels_ref.c) against the torch-operator part
(oracle/vc_oracle.py) and torch's own CPU operators: two independent restatements of the reference's arithmetic must
agree before either is used to judge the HIP kernels.  CPU only.  Bound: fp32 summation order (<= 2e-5 of scale)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import c_kernels, vc_oracle
from openvoice_amd.params import ENC_Q_LAYERS, effective_weight


def _rand(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _close(got, ref, rel=2e-5):
    err = (got - ref).abs().max().item()
    assert err <= rel * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("k,d", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5), (1, 1), (5, 1)])
def test_c_conv1d_matches_torch(k, d):
    """ResBlock1 convs (modules.py:296-306) incl. the leaky-ReLU in front and 'same' padding (commons.py:12-13);
    L shorter than the receptive field of the widest case on purpose (zero padding on both sides at once)."""
    B, cin, cout, L = 2, 6, 5, 37
    x, w, b = _rand(B, cin, L, seed=1), _rand(cout, cin, k, seed=2, scale=(cin * k) ** -0.5), _rand(cout, seed=3)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=(k * d - d) // 2)
    _close(c_kernels.conv1d(x, w, b, dil=d, slope=0.1), ref)
    _close(c_kernels.conv1d(x, w, None, dil=d, slope=1.0), F.conv1d(x, w, None, dilation=d, padding=(k * d - d) // 2))


@pytest.mark.parametrize("k,s", [(16, 8), (4, 2)])
def test_c_conv_transpose1d_matches_torch(k, s):
    """Generator.ups (models.py:244-256): kernel 2 * stride, padding (k - stride) / 2 -> L_out = stride * L."""
    B, cin, cout, L = 2, 6, 4, 11
    x, w, b = _rand(B, cin, L, seed=1), _rand(cin, cout, k, seed=2, scale=0.2), _rand(cout, seed=3)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2)
    got = c_kernels.conv_transpose1d(x, w, b, stride=s, pad=(k - s) // 2, slope=0.1)
    assert got.shape == ref.shape == (B, cout, s * L)
    _close(got, ref)


def test_c_gate_matches_torch():
    B, H, T = 2, 8, 19
    x_in, g = _rand(B, 2 * H, T, seed=1, scale=2.0), _rand(B, 2 * H, seed=2)
    in_act = x_in + g[:, :, None]
    _close(c_kernels.gate(x_in, g), torch.tanh(in_act[:, :H]) * torch.sigmoid(in_act[:, H:]), rel=1e-6)


def test_c_wavenet_matches_python_oracle(synth_sd):
    """The 16-layer posterior-encoder WaveNet (modules.py:185-210) layer by layer in C on the calibrated synthetic
    weights == oracle/vc_oracle.py::wavenet (torch operators), ragged mask included."""
    sd, prefix, B, H, T = synth_sd, "enc_q.enc", 2, 192, 23
    x0 = _rand(B, H, T, seed=5, scale=0.5)
    mask = vc_oracle.sequence_mask(torch.tensor([T, T - 6]), T)
    g = _rand(B, 256, 1, seed=6, scale=0.3)
    with torch.no_grad():
        ref = vc_oracle.wavenet(sd, prefix, x0 * mask, mask, g, ENC_Q_LAYERS)
        g_all = F.conv1d(g, effective_weight(sd, prefix + ".cond_layer"), sd[prefix + ".cond_layer.bias"])[:, :, 0]
        x, out = (x0 * mask).contiguous(), torch.zeros(B, H, T)
        for i in range(ENC_Q_LAYERS):
            c_kernels.wn_layer(x, out, effective_weight(sd, f"{prefix}.in_layers.{i}"), sd[f"{prefix}.in_layers.{i}.bias"],
                               effective_weight(sd, f"{prefix}.res_skip_layers.{i}"),
                               sd[f"{prefix}.res_skip_layers.{i}.bias"],
                               g_all[:, 2 * H * i: 2 * H * (i + 1)].contiguous(), mask[:, 0], 1, i == ENC_Q_LAYERS - 1)
        got = out * mask
    _close(got, ref, rel=5e-5)


def test_c_rq_spline_inverse_matches_torch_oracle_and_inverts_the_forward_spline():
    """transforms.py:50-188 in C == oracle/tts_oracle.py (torch, pinned to the reference's TTS goldens), inside the
    spline, exactly on knots / the tail bound, and in the linear tails."""
    from oracle import tts_oracle
    gen = torch.Generator().manual_seed(11)
    n, nb = 4096, 10
    uw, uh = torch.randn(n, nb, generator=gen), torch.randn(n, nb, generator=gen)
    ud = 2.0 * torch.randn(n, nb - 1, generator=gen)
    y = 7.0 * (2 * torch.rand(n, generator=gen) - 1)            # ~29 % of the elements land in the tails
    y[:4] = torch.tensor([5.0, -5.0, 0.0, 5.0000005])
    ref = tts_oracle.rq_spline_inverse(y, uw, uh, ud)
    got = c_kernels.rq_spline_inverse(y, uw, uh, ud)
    assert torch.equal(got[y.abs() > 5.0], y[y.abs() > 5.0])
    _close(got, ref, rel=3e-5)
    assert bool(((got >= -5.0) & (got <= 5.0))[y.abs() <= 5.0].all())


def test_c_relative_attention_matches_torch_oracle(synth_tts_sd):
    """attentions.py:264-329 in C (band evaluated directly) == oracle/tts_oracle.py::relative_attention (dense [T, T]
    formulation, pinned to the reference's TTS goldens) on the first encoder layer of the synthetic TTS weights,
    ragged lengths, T both below and above the window."""
    from oracle import tts_oracle
    from openvoice_amd.params import ATTN_WINDOW
    sd, prefix, heads = synth_tts_sd, "enc_p.encoder.attn_layers.0", 2
    for T, lens in ((3, [3, 2]), (29, [29, 17])):
        x = _rand(2, 192, T, seed=T, scale=0.7)
        mask = vc_oracle.sequence_mask(torch.tensor(lens), T)
        with torch.no_grad():
            ref = tts_oracle.relative_attention(sd, prefix, x, mask, heads)
            q, k, v = (F.conv1d(x, sd[f"{prefix}.conv_{n}.weight"], sd[f"{prefix}.conv_{n}.bias"]) for n in "qkv")
            core = c_kernels.rel_attention(q, k, v, sd[prefix + ".emb_rel_k"][0], sd[prefix + ".emb_rel_v"][0], mask[:, 0],
                                           heads, ATTN_WINDOW)
            got = F.conv1d(core, sd[prefix + ".conv_o.weight"], sd[prefix + ".conv_o.bias"])
        _close(got, ref, rel=3e-5)
