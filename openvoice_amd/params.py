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
