"""ORACLE (test infrastructure, never the product path).

A plain-PyTorch fp32, CPU restatement of the reference tone-colour-converter inference path.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker; ``openvoice_amd`` never imports ``oracle``.

Parity status: PINNED against the reference itself.  ``oracle/make_golden.py`` imports the
unmodified reference ``SynthesizerTrn`` / ``spectrogram_torch`` from ``/root/reference`` (this
container only), runs them on seeded inputs with the calibrated synthetic weights and commits
the outputs as ``tests/golden/*.pt``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors.  The reference ships no tests or golden vectors of its own (SURVEY.md
section 4), and the released checkpoints are not in the tree, so weights are synthetic.

Each function cites the reference lines it restates.  Weight-norm is folded once
(``openvoice_amd.params.effective_weight``) instead of being re-evaluated per forward; this is
the same expression the reference evaluates (torch ``_weight_norm``), so results agree to fp32
rounding.
"""
import torch
import torch.nn.functional as F

from openvoice_amd.params import (ENC_Q_LAYERS, FLOW_LAYERS, N_FLOWS, REF_ENC_FILTERS,
                                  REF_ENC_GRU, effective_weight)

LRELU_SLOPE = 0.1  # reference: openvoice/modules.py:14


def sequence_mask(lengths, max_length):
    """reference: openvoice/commons.py:121-125 (+ the unsqueeze/cast of models.py:213-215)."""
    t = torch.arange(max_length, dtype=lengths.dtype, device=lengths.device)
    return (t.unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(1).float()


def spectrogram(y, n_fft=1024, hop=256, win=1024):
    """reference: openvoice/mel_processing.py:40-75 -- reflect-pad (n_fft-hop)/2, periodic Hann
    STFT with center=False, magnitude sqrt(re^2 + im^2 + 1e-6)."""
    pad = (n_fft - hop) // 2
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    window = torch.hann_window(win, dtype=y.dtype, device=y.device)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=window, center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    return torch.sqrt(spec.pow(2).sum(-1) + 1e-6)


def wavenet(sd, prefix, x, mask, g, n_layers):
    """WN.forward, reference: openvoice/modules.py:185-210 (gate: openvoice/commons.py:100-107).
    ``g`` is [B or 1, gin, 1]."""
    hidden = x.shape[1]
    out = torch.zeros_like(x)
    g_all = F.conv1d(g, effective_weight(sd, prefix + ".cond_layer"), sd[prefix + ".cond_layer.bias"])
    for i in range(n_layers):
        w_in = effective_weight(sd, f"{prefix}.in_layers.{i}")
        x_in = F.conv1d(x, w_in, sd[f"{prefix}.in_layers.{i}.bias"], padding=(w_in.shape[2] - 1) // 2)
        in_act = x_in + g_all[:, 2 * hidden * i: 2 * hidden * (i + 1)]
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        rs = F.conv1d(acts, effective_weight(sd, f"{prefix}.res_skip_layers.{i}"),
                      sd[f"{prefix}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


def posterior_encoder(sd, spec, mask, g, noise, tau):
    """PosteriorEncoder.forward, reference: openvoice/models.py:212-221.  ``noise`` replaces the
    reference's ``torch.randn_like(m)`` (the only RNG draw on the path) so inputs are identical."""
    x = F.conv1d(spec, sd["enc_q.pre.weight"], sd["enc_q.pre.bias"]) * mask
    x = wavenet(sd, "enc_q.enc", x, mask, g, ENC_Q_LAYERS)
    stats = F.conv1d(x, sd["enc_q.proj.weight"], sd["enc_q.proj.bias"]) * mask
    m, logs = torch.split(stats, stats.shape[1] // 2, dim=1)
    return (m + noise * tau * torch.exp(logs)) * mask


def coupling_layer(sd, prefix, x, mask, g, reverse):
    """ResidualCouplingLayer.forward with mean_only=True, reference: openvoice/modules.py:437-456."""
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = F.conv1d(x0, sd[prefix + ".pre.weight"], sd[prefix + ".pre.bias"]) * mask
    h = wavenet(sd, prefix + ".enc", h, mask, g, FLOW_LAYERS)
    m = F.conv1d(h, sd[prefix + ".post.weight"], sd[prefix + ".post.bias"]) * mask
    if not reverse:
        x1 = m + x1 * mask            # exp(logs) == 1 (mean_only)
    else:
        x1 = (x1 - m) * mask
    return torch.cat([x0, x1], 1)


def flow(sd, x, mask, g, reverse):
    """ResidualCouplingBlock.forward, reference: openvoice/models.py:390-397; Flip:
    openvoice/modules.py:374-381."""
    if not reverse:
        for f in range(N_FLOWS):
            x = coupling_layer(sd, f"flow.flows.{2 * f}", x, mask, g, False)
            x = torch.flip(x, [1])
    else:
        for f in reversed(range(N_FLOWS)):
            x = torch.flip(x, [1])
            x = coupling_layer(sd, f"flow.flows.{2 * f}", x, mask, g, True)
    return x


def resblock1(sd, prefix, x, kernel, dilations):
    """ResBlock1.forward with x_mask=None, reference: openvoice/modules.py:296-309."""
    for n, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, effective_weight(sd, f"{prefix}.convs1.{n}"), sd[f"{prefix}.convs1.{n}.bias"],
                      dilation=d, padding=(kernel * d - d) // 2)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, effective_weight(sd, f"{prefix}.convs2.{n}"), sd[f"{prefix}.convs2.{n}.bias"],
                      padding=(kernel - 1) // 2)
        x = xt + x
    return x


def generator(sd, z, g, cfg):
    """Generator.forward, reference: openvoice/models.py:272-291.  Note the final leaky_relu uses
    torch's default slope 0.01 (models.py:287), not LRELU_SLOPE."""
    x = F.conv1d(z, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, sd["dec.cond.weight"], sd["dec.cond.bias"])
    kernels, dils = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, effective_weight(sd, f"dec.ups.{i}"), sd[f"dec.ups.{i}.bias"],
                               stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(kernels, dils)):
            y = resblock1(sd, f"dec.resblocks.{i * len(kernels) + j}", x, rk, rd)
            xs = y if xs is None else xs + y
        x = xs / len(kernels)
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


def voice_conversion(sd, cfg, spec, spec_lengths, sid_src, sid_tgt, tau, noise, zero_g=False):
    """SynthesizerTrn.voice_conversion, reference: openvoice/models.py:492-499.
    Returns ``(o_hat, y_mask, (z, z_p, z_hat))`` like the reference."""
    mask = sequence_mask(spec_lengths, spec.shape[2])
    g_src, g_tgt = sid_src, sid_tgt
    z = posterior_encoder(sd, spec, mask, torch.zeros_like(g_src) if zero_g else g_src, noise, tau)
    z_p = flow(sd, z, mask, g_src, False)
    z_hat = flow(sd, z_p, mask, g_tgt, True)
    o_hat = generator(sd, z_hat * mask, torch.zeros_like(g_tgt) if zero_g else g_tgt, cfg)
    return o_hat, mask, (z, z_p, z_hat)


def reference_encoder(sd, spec_t):
    """ReferenceEncoder.forward, reference: openvoice/models.py:339-359.  ``spec_t`` is
    [N, Ty, n_freq]; returns [N, gin]."""
    n, ty, nf = spec_t.shape
    out = F.layer_norm(spec_t.reshape(n, 1, ty, nf), (nf,), sd["ref_enc.layernorm.weight"],
                       sd["ref_enc.layernorm.bias"])
    for i in range(len(REF_ENC_FILTERS)):
        out = F.relu(F.conv2d(out, effective_weight(sd, f"ref_enc.convs.{i}"),
                              sd[f"ref_enc.convs.{i}.bias"], stride=2, padding=1))
    out = out.transpose(1, 2)
    t2 = out.shape[1]
    out = out.contiguous().view(n, t2, -1)
    w_ih, w_hh = sd["ref_enc.gru.weight_ih_l0"], sd["ref_enc.gru.weight_hh_l0"]
    b_ih, b_hh = sd["ref_enc.gru.bias_ih_l0"], sd["ref_enc.gru.bias_hh_l0"]
    h = torch.zeros(n, REF_ENC_GRU, dtype=out.dtype, device=out.device)
    for t in range(t2):  # torch.nn.GRU cell equations (gate order r, z, n)
        gi = out[:, t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        zg = torch.sigmoid(i_z + h_z)
        ng = torch.tanh(i_n + r * h_n)
        h = (1 - zg) * ng + zg * h
    return h @ sd["ref_enc.proj.weight"].t() + sd["ref_enc.proj.bias"]
