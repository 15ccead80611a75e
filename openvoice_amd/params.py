"""Checkpoint schema of the tone-colour converter and helpers to read it.

The released converter ``checkpoint.pth`` holds ``{'model': state_dict}`` whose keys are the
parameter names of the reference ``SynthesizerTrn`` built with ``n_speakers == 0``
(reference: openvoice/models.py:399-465; load path openvoice/api.py:35-39).  Weight-normed
layers are stored as ``weight_g`` / ``weight_v`` pairs because the reference never calls
``remove_weight_norm``.  This module

* enumerates that schema from the config (``converter_param_spec``) so the engine can own
  same-named parameters without containing any reference module code,
* folds weight-norm at load time (``effective_weight``): ``w = g * v / ||v||`` with the norm
  over every dim except 0 -- for ``ConvTranspose1d`` dim 0 is C_in,
* generates the calibrated synthetic weight set used when no checkpoint exists
  (``synthetic_state_dict``; recipe: SURVEY.md Appendix B).
"""
from collections import OrderedDict

import torch

# Hard-wired in the reference code, not in config.json (reference: openvoice/models.py:442-448,
# :374, :310).
ENC_Q_KERNEL = 5
ENC_Q_LAYERS = 16
FLOW_KERNEL = 5
FLOW_LAYERS = 4
N_FLOWS = 4
REF_ENC_FILTERS = (32, 32, 64, 64, 128, 128)
REF_ENC_GRU = 128


def _wn(spec, prefix, shape_v, bias_len):
    """weight-normed layer: bias, weight_g (dim-0 sized, ones elsewhere), weight_v."""
    spec[prefix + ".bias"] = (bias_len,)
    spec[prefix + ".weight_g"] = (shape_v[0],) + (1,) * (len(shape_v) - 1)
    spec[prefix + ".weight_v"] = tuple(shape_v)


def _wavenet(spec, prefix, hidden, kernel, n_layers, gin):
    """WN stack parameters (reference: openvoice/modules.py:133-183)."""
    for i in range(n_layers):
        _wn(spec, f"{prefix}.in_layers.{i}", (2 * hidden, hidden, kernel), 2 * hidden)
    for i in range(n_layers):
        rs = 2 * hidden if i < n_layers - 1 else hidden
        _wn(spec, f"{prefix}.res_skip_layers.{i}", (rs, hidden, 1), rs)
    _wn(spec, f"{prefix}.cond_layer", (2 * hidden * n_layers, gin, 1), 2 * hidden * n_layers)


def ref_enc_out_bins(spec_channels, n_convs=len(REF_ENC_FILTERS)):
    """Frequency bins left after the stride-2 stack (reference: openvoice/models.py:361-364)."""
    bins = spec_channels
    for _ in range(n_convs):
        bins = (bins - 3 + 2) // 2 + 1
    return bins


def converter_param_spec(spec_channels, inter_channels, hidden_channels, resblock,
                         resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                         upsample_initial_channel, upsample_kernel_sizes, gin_channels=256,
                         **_unused):
    """Ordered ``{name: shape}`` of every tensor in the converter state dict."""
    if str(resblock) != "1":
        raise NotImplementedError("only resblock '1' (ResBlock1) is used by the released converters")
    spec = OrderedDict()
    # dec = HiFi-GAN Generator (reference: openvoice/models.py:225-270)
    spec["dec.conv_pre.weight"] = (upsample_initial_channel, inter_channels, 7)
    spec["dec.conv_pre.bias"] = (upsample_initial_channel,)
    ch = upsample_initial_channel
    for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
        _wn(spec, f"dec.ups.{i}", (ch, ch // 2, k), ch // 2)
        ch //= 2
    ch = upsample_initial_channel
    for i in range(len(upsample_rates)):
        ch //= 2
        for j, (k, dil) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            rb = f"dec.resblocks.{i * len(resblock_kernel_sizes) + j}"
            for n in range(len(dil)):
                _wn(spec, f"{rb}.convs1.{n}", (ch, ch, k), ch)
            for n in range(len(dil)):
                _wn(spec, f"{rb}.convs2.{n}", (ch, ch, k), ch)
    spec["dec.conv_post.weight"] = (1, ch, 7)
    spec["dec.cond.weight"] = (upsample_initial_channel, gin_channels, 1)
    spec["dec.cond.bias"] = (upsample_initial_channel,)
    # enc_q = PosteriorEncoder (reference: openvoice/models.py:182-210)
    spec["enc_q.pre.weight"] = (hidden_channels, spec_channels, 1)
    spec["enc_q.pre.bias"] = (hidden_channels,)
    _wavenet(spec, "enc_q.enc", hidden_channels, ENC_Q_KERNEL, ENC_Q_LAYERS, gin_channels)
    spec["enc_q.proj.weight"] = (2 * inter_channels, hidden_channels, 1)
    spec["enc_q.proj.bias"] = (2 * inter_channels,)
    # flow = ResidualCouplingBlock; Flip modules (odd indices) own no parameters
    # (reference: openvoice/models.py:367-388, openvoice/modules.py:402-435)
    half = inter_channels // 2
    for f in range(N_FLOWS):
        p = f"flow.flows.{2 * f}"
        spec[p + ".pre.weight"] = (hidden_channels, half, 1)
        spec[p + ".pre.bias"] = (hidden_channels,)
        _wavenet(spec, p + ".enc", hidden_channels, FLOW_KERNEL, FLOW_LAYERS, gin_channels)
        spec[p + ".post.weight"] = (half, hidden_channels, 1)
        spec[p + ".post.bias"] = (half,)
    # ref_enc = ReferenceEncoder (reference: openvoice/models.py:307-337)
    filters = (1,) + REF_ENC_FILTERS
    for i in range(len(REF_ENC_FILTERS)):
        _wn(spec, f"ref_enc.convs.{i}", (filters[i + 1], filters[i], 3, 3), filters[i + 1])
    gru_in = REF_ENC_FILTERS[-1] * ref_enc_out_bins(spec_channels)
    spec["ref_enc.gru.weight_ih_l0"] = (3 * REF_ENC_GRU, gru_in)
    spec["ref_enc.gru.weight_hh_l0"] = (3 * REF_ENC_GRU, REF_ENC_GRU)
    spec["ref_enc.gru.bias_ih_l0"] = (3 * REF_ENC_GRU,)
    spec["ref_enc.gru.bias_hh_l0"] = (3 * REF_ENC_GRU,)
    spec["ref_enc.proj.weight"] = (gin_channels, REF_ENC_GRU)
    spec["ref_enc.proj.bias"] = (gin_channels,)
    spec["ref_enc.layernorm.weight"] = (spec_channels,)
    spec["ref_enc.layernorm.bias"] = (spec_channels,)
    return spec


# ---- V1 base-speaker TTS (n_speakers > 0): text encoder, duration predictors, speaker table ----------
SDP_FILTER = 192          # StochasticDurationPredictor(hidden, 192, 3, 0.5, 4) -- reference: openvoice/models.py:461
SDP_KERNEL = 3
SDP_FLOWS = 4
SDP_DDS_LAYERS = 3
SDP_NUM_BINS = 10         # ConvFlow defaults, reference: openvoice/modules.py:466-467
SDP_TAIL_BOUND = 5.0
DP_FILTER = 256           # DurationPredictor(hidden, 256, 3, 0.5) -- reference: openvoice/models.py:462
DP_KERNEL = 3
ATTN_WINDOW = 4           # attentions.Encoder default window_size, reference: openvoice/attentions.py:46


def _dds(spec, prefix, channels, kernel, n_layers):
    """DDSConv parameters (reference: openvoice/modules.py:84-115)."""
    for i in range(n_layers):
        spec[f"{prefix}.convs_sep.{i}.weight"] = (channels, 1, kernel)
        spec[f"{prefix}.convs_sep.{i}.bias"] = (channels,)
    for i in range(n_layers):
        spec[f"{prefix}.convs_1x1.{i}.weight"] = (channels, channels, 1)
        spec[f"{prefix}.convs_1x1.{i}.bias"] = (channels,)
    for name in ("norms_1", "norms_2"):
        for i in range(n_layers):
            spec[f"{prefix}.{name}.{i}.gamma"] = (channels,)
            spec[f"{prefix}.{name}.{i}.beta"] = (channels,)


def _sdp_flows(spec, prefix, filt):
    """ElementwiseAffine(2) + 4 x (ConvFlow(2, filt, 3, n_layers=3), Flip) -- reference:
    openvoice/models.py:113-127; Flip (even indices 2, 4, ..) owns no parameters."""
    spec[f"{prefix}.0.m"] = (2, 1)
    spec[f"{prefix}.0.logs"] = (2, 1)
    for f in range(SDP_FLOWS):
        p = f"{prefix}.{2 * f + 1}"
        spec[p + ".pre.weight"] = (filt, 1, 1)
        spec[p + ".pre.bias"] = (filt,)
        _dds(spec, p + ".convs", filt, SDP_KERNEL, SDP_DDS_LAYERS)
        spec[p + ".proj.weight"] = (3 * SDP_NUM_BINS - 1, filt, 1)
        spec[p + ".proj.bias"] = (3 * SDP_NUM_BINS - 1,)


def tts_param_spec(n_vocab, n_speakers, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                   kernel_size, gin_channels=256, **_unused):
    """Ordered ``{name: shape}`` of the tensors the V1 TTS model adds to the converter schema
    (``enc_p``, ``sdp``, ``dp``, ``emb_g``; reference: openvoice/models.py:16-180, :450-464,
    openvoice/attentions.py:37-121, :210-262, :410-436)."""
    spec = OrderedDict()
    H, dk = hidden_channels, hidden_channels // n_heads
    spec["enc_p.emb.weight"] = (n_vocab, H)
    for i in range(n_layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        spec[a + ".emb_rel_k"] = (1, 2 * ATTN_WINDOW + 1, dk)
        spec[a + ".emb_rel_v"] = (1, 2 * ATTN_WINDOW + 1, dk)
        for c in ("conv_q", "conv_k", "conv_v", "conv_o"):
            spec[f"{a}.{c}.weight"] = (H, H, 1)
            spec[f"{a}.{c}.bias"] = (H,)
    for i in range(n_layers):
        spec[f"enc_p.encoder.norm_layers_1.{i}.gamma"] = (H,)
        spec[f"enc_p.encoder.norm_layers_1.{i}.beta"] = (H,)
    for i in range(n_layers):
        f = f"enc_p.encoder.ffn_layers.{i}"
        spec[f + ".conv_1.weight"] = (filter_channels, H, kernel_size)
        spec[f + ".conv_1.bias"] = (filter_channels,)
        spec[f + ".conv_2.weight"] = (H, filter_channels, kernel_size)
        spec[f + ".conv_2.bias"] = (H,)
    for i in range(n_layers):
        spec[f"enc_p.encoder.norm_layers_2.{i}.gamma"] = (H,)
        spec[f"enc_p.encoder.norm_layers_2.{i}.beta"] = (H,)
    spec["enc_p.proj.weight"] = (2 * inter_channels, H, 1)
    spec["enc_p.proj.bias"] = (2 * inter_channels,)
    F = SDP_FILTER
    _sdp_flows(spec, "sdp.flows", F)
    spec["sdp.post_pre.weight"] = (F, 1, 1)
    spec["sdp.post_pre.bias"] = (F,)
    spec["sdp.post_proj.weight"] = (F, F, 1)
    spec["sdp.post_proj.bias"] = (F,)
    _dds(spec, "sdp.post_convs", F, SDP_KERNEL, SDP_DDS_LAYERS)
    _sdp_flows(spec, "sdp.post_flows", F)
    spec["sdp.pre.weight"] = (F, H, 1)
    spec["sdp.pre.bias"] = (F,)
    spec["sdp.proj.weight"] = (F, F, 1)
    spec["sdp.proj.bias"] = (F,)
    _dds(spec, "sdp.convs", F, SDP_KERNEL, SDP_DDS_LAYERS)
    spec["sdp.cond.weight"] = (F, gin_channels, 1)
    spec["sdp.cond.bias"] = (F,)
    spec["dp.conv_1.weight"] = (DP_FILTER, H, DP_KERNEL)
    spec["dp.conv_1.bias"] = (DP_FILTER,)
    spec["dp.norm_1.gamma"] = (DP_FILTER,)
    spec["dp.norm_1.beta"] = (DP_FILTER,)
    spec["dp.conv_2.weight"] = (DP_FILTER, DP_FILTER, DP_KERNEL)
    spec["dp.conv_2.bias"] = (DP_FILTER,)
    spec["dp.norm_2.gamma"] = (DP_FILTER,)
    spec["dp.norm_2.beta"] = (DP_FILTER,)
    spec["dp.proj.weight"] = (1, DP_FILTER, 1)
    spec["dp.proj.bias"] = (1,)
    spec["dp.cond.weight"] = (H, gin_channels, 1)
    spec["dp.cond.bias"] = (H,)
    spec["emb_g.weight"] = (n_speakers, gin_channels)
    return spec


def tts_full_param_spec(n_vocab, n_speakers, spec_channels, **cfg):
    """Schema of ``SynthesizerTrn(n_vocab, spec_channels, n_speakers > 0)``: dec, enc_q, flow, then
    enc_p / sdp / dp / emb_g (no ref_enc) -- reference: openvoice/models.py:428-464."""
    spec = OrderedDict((k, v) for k, v in converter_param_spec(spec_channels, **cfg).items()
                       if not k.startswith("ref_enc."))
    spec.update(tts_param_spec(n_vocab, n_speakers, **cfg))
    return spec


def synthetic_tts_state_dict(hps_model, n_vocab=68, n_speakers=10, spec_channels=513, seed=4321):
    """Calibrated random weights for the V1 TTS model: the converter recipe for dec/enc_q/flow plus
    fan-in scaled text-encoder / duration-predictor weights chosen so that predicted durations are a
    few frames per token (not all 1, not hundreds) and the spline flows are non-trivial."""
    cfg = dict(hps_model.items()) if hasattr(hps_model, "items") else dict(hps_model)
    base = synthetic_state_dict(cfg, spec_channels, seed=seed)
    out = OrderedDict((k, v) for k, v in base.items() if not k.startswith("ref_enc."))
    gen = torch.Generator().manual_seed(seed + 1)
    for name, shape in tts_param_spec(n_vocab, n_speakers, **cfg).items():
        if name.endswith(".gamma"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
        elif name.endswith(".beta") or (name.endswith(".bias") and len(shape) == 1):
            t = 0.05 * torch.randn(shape, generator=gen)
        elif name == "enc_p.emb.weight":
            t = torch.randn(shape, generator=gen) * shape[1] ** -0.5
        elif name == "emb_g.weight":
            t = 0.3 * torch.randn(shape, generator=gen)
        elif name.endswith(("emb_rel_k", "emb_rel_v")):
            t = torch.randn(shape, generator=gen) * shape[2] ** -0.5
        elif name.endswith((".m", ".logs")):
            t = 0.2 * torch.randn(shape, generator=gen)
        else:
            fan = 1
            for s in shape[1:]:
                fan *= s
            gain = 1.0
            if ".convs_sep." in name:
                gain = 1.5
            if name.endswith("proj.weight") and ".flows." in name:
                gain = 3.0          # spline parameters: make bins/derivatives visibly non-uniform
            t = gain * torch.randn(shape, generator=gen) / fan ** 0.5
        out[name] = t
    # durations: logw ~ 0.2 * sdp + 0.8 * dp; aim at exp(logw) ~ 2-6 frames per token
    out["dp.proj.bias"] = torch.tensor([1.2])
    out["dp.proj.weight"] = out["dp.proj.weight"] * 0.5
    return out


def effective_weight(sd, prefix):
    """Dense weight of layer ``prefix``: plain ``.weight`` or folded ``weight_g``/``weight_v``.

    ``torch.nn.utils.weight_norm(dim=0)`` computes ``g * v / ||v||`` with the 2-norm over all
    dims except 0 on every forward (reference re-evaluates it 181x per conversion, SURVEY.md
    section 2a); folding once at load is exact up to fp32 rounding of the same expression.
    """
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].detach().float()
    v = sd[prefix + ".weight_v"].detach().float()
    g = sd[prefix + ".weight_g"].detach().float()
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / norm)


def stress_state_dict(sd, gain=4.0):
    """High-dynamic-range variant of a synthetic converter state dict (parity stress case):
    the posterior mean head and every coupling's ``post`` are scaled by ``gain`` -- latents ``z`` / ``z_p`` and the flow
    shifts grow ``gain``-fold, so the forward/reverse flow cancellation runs at ``gain`` times the magnitude -- and
    ``dec.conv_pre`` by 1/``gain`` so the generator still sees an O(1) input and the waveform stays unsaturated (an
    error would otherwise hide behind tanh)."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    inter = out["enc_q.proj.weight"].shape[0] // 2
    out["enc_q.proj.weight"][:inter] *= gain
    out["enc_q.proj.bias"][:inter] *= gain
    for k in out:
        if k.startswith("flow.flows.") and (k.endswith(".post.weight") or k.endswith(".post.bias")):
            out[k] *= gain
    for k in ("dec.conv_pre.weight_g", "dec.conv_pre.weight"):
        if k in out:
            out[k] = out[k] / gain
    return out


def synthetic_state_dict(hps_model, spec_channels=513, seed=1234, perturb_g=True):
    """Calibrated random converter weights.

    Default initialisation makes parity vacuous (zero-initialised coupling ``post`` => the flow
    is the identity; decoder weights ~N(0, 0.01) => |o_hat| ~ 0.03), so tests and ``bench.py``
    use this recipe instead: O(1) activations everywhere, a non-trivial flow and an unsaturated
    waveform.  ``perturb_g`` scales every ``weight_g`` by U(0.5, 1.5) so that the weight-norm
    folding path (incl. the ConvTranspose axis) is exercised.
    """
    spec = converter_param_spec(spec_channels, **dict(hps_model.items()) if hasattr(hps_model, "items") else hps_model)
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in spec.items():
        if name.endswith("weight_g"):
            continue
        if name.startswith("ref_enc.layernorm"):
            sd[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            continue
        if len(shape) < 2:
            sd[name] = 0.02 * torch.randn(shape, generator=gen)
            continue
        if name.startswith("dec.ups."):
            fan = 2 * shape[0]          # K/stride = 2 taps of every C_in reach each output
        else:
            fan = 1
            for s in shape[1:]:
                fan *= s
        gain = 1.0
        if ".res_skip_layers." in name or ".convs2." in name or "cond" in name:
            gain = 0.5
        sd[name] = gain * torch.randn(shape, generator=gen) / fan ** 0.5
    inter = spec["enc_q.proj.weight"][0] // 2
    sd["enc_q.proj.weight"][inter:] *= 0.05       # small posterior log-std head
    sd["enc_q.proj.bias"][inter:] = -1.0
    out = OrderedDict()
    for name, shape in spec.items():
        if name.endswith("weight_g"):
            v = sd[name[:-1] + "v"]
            g = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape)
            if perturb_g:
                g = g * (0.5 + torch.rand(shape, generator=gen))
            out[name] = g
        else:
            out[name] = sd[name]
    return out
