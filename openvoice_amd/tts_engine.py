"""Host-side driver of the V1 base-speaker TTS model, ``SynthesizerTrn.infer``
(reference: openvoice/models.py:467-490).

The token-rate front end -- text encoder with windowed relative-position attention
(models.py:48-57, attentions.py:104-121, :264-329, :440-448), the deterministic and the stochastic
duration predictor (models.py:86-100, :129-138 / :171-180 with DDSConv modules.py:117-130 and the
inverse rational-quadratic spline of ConvFlow modules.py:485-516 / transforms.py:50-188), the
duration arithmetic and the duration-driven expansion of the prior (models.py:474-487) -- runs on the
kernels of csrc/tts.hip plus the MFMA conv kernel for every dense 1x1 / k3 conv.  The frame-rate back
end (flow in reverse, HiFi-GAN generator: 98 % of the FLOPs) is the converter engine's
(``engine.ConverterEngine._flow`` / ``.decode``), conditioned on the speaker-table embedding.

Load-time transforms: the q, k, v projections of a layer are concatenated into one 576-row conv; the
single-row ``dp.proj`` and the 29-row spline projections are zero-padded to 32 rows (the conv kernel
stores whole 32-row fragments); the Flips between the spline flows are not materialised (channel roles
alternate); of ``sdp.flows.0`` (ElementwiseAffine) only channel 0 is ever read, so it is two scalars.
"""
import math

import torch

from . import _lib
from ._lib import F_MASK_V, LN_POST_GELU, LN_PRE_RELU
from .engine import ConverterEngine, PackedConv, on_own_device, padded_frames
from .params import ATTN_WINDOW, SDP_DDS_LAYERS, SDP_FLOWS, SDP_KERNEL, SDP_NUM_BINS, SDP_TAIL_BOUND

LN_EPS = 1e-5   # modules.LayerNorm / attentions.LayerNorm default


def _pad_rows(w, b, rows):
    """Zero-pad a [r, Cin, K] weight and its bias to ``rows`` output rows."""
    wp = torch.zeros(rows, w.shape[1], w.shape[2], dtype=torch.float32)
    wp[:w.shape[0]] = w
    bp = torch.zeros(rows, dtype=torch.float32)
    bp[:b.shape[0]] = b
    return wp, bp


class _DDS:
    """Packed DDSConv (reference: openvoice/modules.py:84-130)."""

    def __init__(self, sd, prefix, device):
        self.layers = []
        for i in range(SDP_DDS_LAYERS):
            self.layers.append(dict(
                dil=SDP_KERNEL ** i,
                w_sep=sd[f"{prefix}.convs_sep.{i}.weight"][:, 0].contiguous().to(device),
                b_sep=sd[f"{prefix}.convs_sep.{i}.bias"].contiguous().to(device),
                c1x1=PackedConv(sd[f"{prefix}.convs_1x1.{i}.weight"], sd[f"{prefix}.convs_1x1.{i}.bias"], device, K=1),
                g1=sd[f"{prefix}.norms_1.{i}.gamma"].to(device), b1=sd[f"{prefix}.norms_1.{i}.beta"].to(device),
                g2=sd[f"{prefix}.norms_2.{i}.gamma"].to(device), b2=sd[f"{prefix}.norms_2.{i}.beta"].to(device)))


class TtsEngine:
    """Kernel-level implementation of the V1 TTS model for one device."""

    def __init__(self, state_dict, model_cfg, spec_channels, device, n_vocab, n_speakers):
        self.device = torch.device(device)
        cfg = dict(model_cfg.items()) if hasattr(model_cfg, "items") else dict(model_cfg)
        self.cfg = cfg
        self.core = ConverterEngine(state_dict, cfg, spec_channels, device, zero_g=False)
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        dev = self.device
        self.n_vocab, self.n_speakers = n_vocab, n_speakers
        H = self.H = cfg["hidden_channels"]
        self.inter = cfg["inter_channels"]
        self.n_heads, self.n_layers, ks = cfg["n_heads"], cfg["n_layers"], cfg["kernel_size"]
        assert ks % 2 == 1, "FFN 'same' padding is symmetric only for odd kernels (attentions.py:457-465)"
        self.emb = sd["enc_p.emb.weight"].contiguous().to(dev)
        self.layers = []
        for i in range(self.n_layers):
            a, f = f"enc_p.encoder.attn_layers.{i}", f"enc_p.encoder.ffn_layers.{i}"
            wqkv = torch.cat([sd[f"{a}.conv_{c}.weight"] for c in "qkv"], 0)
            bqkv = torch.cat([sd[f"{a}.conv_{c}.bias"] for c in "qkv"], 0)
            self.layers.append(dict(
                qkv=PackedConv(wqkv, bqkv, dev, K=1),
                o=PackedConv(sd[a + ".conv_o.weight"], sd[a + ".conv_o.bias"], dev, K=1),
                emb_k=sd[a + ".emb_rel_k"][0].contiguous().to(dev), emb_v=sd[a + ".emb_rel_v"][0].contiguous().to(dev),
                g1=sd[f"enc_p.encoder.norm_layers_1.{i}.gamma"].to(dev), b1=sd[f"enc_p.encoder.norm_layers_1.{i}.beta"].to(dev),
                ffn1=PackedConv(sd[f + ".conv_1.weight"], sd[f + ".conv_1.bias"], dev, K=ks),
                ffn2=PackedConv(sd[f + ".conv_2.weight"], sd[f + ".conv_2.bias"], dev, K=ks),
                g2=sd[f"enc_p.encoder.norm_layers_2.{i}.gamma"].to(dev), b2=sd[f"enc_p.encoder.norm_layers_2.{i}.beta"].to(dev)))
        self.proj = PackedConv(sd["enc_p.proj.weight"], sd["enc_p.proj.bias"], dev, K=1)
        # duration predictor
        self.dp_cond_w = sd["dp.cond.weight"][:, :, 0].contiguous().to(dev)
        self.dp_cond_b = sd["dp.cond.bias"].contiguous().to(dev)
        self.dp_c1 = PackedConv(sd["dp.conv_1.weight"], sd["dp.conv_1.bias"], dev, K=sd["dp.conv_1.weight"].shape[2])
        self.dp_c2 = PackedConv(sd["dp.conv_2.weight"], sd["dp.conv_2.bias"], dev, K=sd["dp.conv_2.weight"].shape[2])
        self.dp_n = [(sd[f"dp.norm_{n}.gamma"].to(dev), sd[f"dp.norm_{n}.beta"].to(dev)) for n in (1, 2)]
        self.dp_filter = sd["dp.conv_1.weight"].shape[0]
        wp, bp = _pad_rows(sd["dp.proj.weight"], sd["dp.proj.bias"], 32)
        self.dp_proj = PackedConv(wp, bp, dev, K=1)
        # stochastic duration predictor (reverse direction only)
        self.sdp_filter = F = sd["sdp.pre.weight"].shape[0]
        self.sdp_pre = PackedConv(sd["sdp.pre.weight"], sd["sdp.pre.bias"], dev, K=1)
        self.sdp_cond_w = sd["sdp.cond.weight"][:, :, 0].contiguous().to(dev)
        self.sdp_cond_b = sd["sdp.cond.bias"].contiguous().to(dev)
        self.sdp_convs = _DDS(sd, "sdp.convs", dev)
        self.sdp_proj = PackedConv(sd["sdp.proj.weight"], sd["sdp.proj.bias"], dev, K=1)
        self.sdp_flows = []
        for f in range(SDP_FLOWS, 1, -1):          # reverse order; ConvFlow 1 is dropped (models.py:173)
            p = f"sdp.flows.{2 * f - 1}"
            wp, bp = _pad_rows(sd[p + ".proj.weight"], sd[p + ".proj.bias"], 32)
            self.sdp_flows.append(dict(pre_w=sd[p + ".pre.weight"][:, 0, 0].contiguous().to(dev),
                                       pre_b=sd[p + ".pre.bias"].contiguous().to(dev),
                                       dds=_DDS(sd, p + ".convs", dev), proj=PackedConv(wp, bp, dev, K=1)))
        self.ea_m = float(sd["sdp.flows.0.m"][0, 0])
        self.ea_logs = float(sd["sdp.flows.0.logs"][0, 0])
        self.emb_g = sd["emb_g.weight"].contiguous().to(dev)

    # ---- helpers ---------------------------------------------------------------------------------------
    def _conv(self, layer, x, xc, out, oc, B, T, ld, **kw):
        """Token-rate conv: x (B, xc, ld) -> out (B, oc, ld)."""
        self.core._conv(layer, x, 0, xc * ld, out, 0, oc * ld, B, T, x_ld=ld, out_ld=ld, tag="tts", **kw)

    def _ln(self, x, gamma, beta, out, B, C, T, ld, res=None, res2=None, mask=None, flags=0):
        _lib.call("ov_layernorm_ch_f32", x, res, gamma, beta, res2, mask, out, B, C, T, ld, LN_EPS, flags)

    def _dds(self, dds, x, tmp1, tmp2, mask, B, C, T, ld):
        """DDSConv in place on ``x`` (modules.py:117-130); ``x`` already holds ``x + g`` when conditioned."""
        n = len(dds.layers)
        for i, L in enumerate(dds.layers):
            _lib.call("ov_dwconv1d_f32", x, L["w_sep"], L["b_sep"], mask, tmp1, B, C, T, ld, SDP_KERNEL, L["dil"])
            self._ln(tmp1, L["g1"], L["b1"], tmp1, B, C, T, ld, flags=LN_POST_GELU)
            self._conv(L["c1x1"], tmp1, C, tmp2, C, B, T, ld)
            # x = x + gelu(LN(y)); the block's final `x * mask` rides on the last layer
            self._ln(tmp2, L["g2"], L["b2"], x, B, C, T, ld, res2=x, mask=mask if i == n - 1 else None,
                     flags=LN_POST_GELU)

    # ---- the path ----------------------------------------------------------------------------------------
    @torch.no_grad()
    @on_own_device
    def infer(self, tokens, lengths, sid, noise_scale=1.0, length_scale=1.0, noise_scale_w=1.0, sdp_ratio=0.2,
              max_len=None, noise_w=None, noise_z=None, return_attn=True, skip_padding=False):
        """Same contract as the reference (models.py:467-490): returns
        ``(o [B,1,256*Ty'], attn [B,1,Ty,Tx], y_mask [B,1,Ty], (z, z_p, m_p, logs_p) [B,192,Ty])``.
        ``noise_w`` [B,2,Tx] / ``noise_z`` [B,192,>=Ty] replace the reference's two RNG draws when given.
        ``skip_padding``: the generator computes only ``y_length + 16`` frames of each utterance of a padded batch
        (valid samples bit-identical, the padded tail of ``o`` zero; ``ConverterEngine.voice_conversion``)."""
        dev = self.device
        tokens = tokens.to(dev, torch.int64).contiguous()
        lengths = lengths.to(dev, torch.int64).contiguous()
        sid = sid.to(dev, torch.int64).reshape(-1)
        B, Tx = tokens.shape
        if int(tokens.min()) < 0 or int(tokens.max()) >= self.n_vocab:
            raise IndexError("token id out of range")            # nn.Embedding's error in the reference
        if int(sid.min()) < 0 or int(sid.max()) >= self.n_speakers:
            raise IndexError("speaker id out of range")
        H, C, Lx = self.H, self.inter, padded_frames(Tx)
        f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        x, y, mask = f(B, H, Lx), f(B, H, Lx), f(B, Lx)
        _lib.call("ov_sequence_mask_f32", lengths, mask, B, Tx, Lx)
        _lib.call("ov_embed_f32", tokens, self.emb, lengths, x, B, Tx, H, self.n_vocab, Lx, math.sqrt(H))
        # ---- text encoder (attentions.py:104-121) ---------------------------------------------------------
        qkv, att = f(B, 3 * H, Lx), f(B, H, Lx)
        filt = self.layers[0]["ffn1"].rows
        hid = f(B, filt, Lx)
        dk = H // self.n_heads
        for L in self.layers:
            self._conv(L["qkv"], x, H, qkv, 3 * H, B, Tx, Lx)
            _lib.call("ov_rel_attention_f32", qkv, (qkv, H * Lx), (qkv, 2 * H * Lx), L["emb_k"], L["emb_v"], mask, att,
                      3 * H * Lx, H * Lx, B, self.n_heads, dk, Tx, Lx, ATTN_WINDOW)
            self._conv(L["o"], att, H, y, H, B, Tx, Lx)
            self._ln(x, L["g1"], L["b1"], x, B, H, Tx, Lx, res=y, mask=mask)
            self._conv(L["ffn1"], x, H, hid, filt, B, Tx, Lx, flags=F_MASK_V, mask=mask, mask_bs=Lx)
            self._conv(L["ffn2"], hid, filt, y, H, B, Tx, Lx, in_slope=0.0, flags=F_MASK_V, mask=mask, mask_bs=Lx)
            self._ln(x, L["g2"], L["b2"], x, B, H, Tx, Lx, res=y, mask=mask)
        stats = f(B, 2 * C, Lx)
        self._conv(self.proj, x, H, stats, 2 * C, B, Tx, Lx, flags=F_MASK_V, mask=mask, mask_bs=Lx)
        g = self.emb_g.index_select(0, sid).contiguous()                       # [B, gin]  (models.py:470)
        # ---- duration predictor (models.py:86-100) --------------------------------------------------------
        Fd = self.dp_filter
        xd, d1, d2, dp_out = f(B, H, Lx), f(B, Fd, Lx), f(B, Fd, Lx), f(B, 32, Lx)
        cg = self.core._linear(g, self.dp_cond_w, self.dp_cond_b)
        _lib.call("ov_add_bias_mask_f32", x, cg, mask, xd, B, H, Tx, Lx)
        self._conv(self.dp_c1, xd, H, d1, Fd, B, Tx, Lx)
        self._ln(d1, *self.dp_n[0], d1, B, Fd, Tx, Lx, mask=mask, flags=LN_PRE_RELU)
        self._conv(self.dp_c2, d1, Fd, d2, Fd, B, Tx, Lx)
        self._ln(d2, *self.dp_n[1], d2, B, Fd, Tx, Lx, mask=mask, flags=LN_PRE_RELU)
        self._conv(self.dp_proj, d2, Fd, dp_out, 32, B, Tx, Lx, flags=F_MASK_V, mask=mask, mask_bs=Lx)
        # ---- stochastic duration predictor, reverse (models.py:129-138, :171-180) -------------------------
        Fs = self.sdp_filter
        xs, t1, t2, hflow, hp = f(B, Fs, Lx), f(B, Fs, Lx), f(B, Fs, Lx), f(B, Fs, Lx), f(B, 32, Lx)
        cs = self.core._linear(g, self.sdp_cond_w, self.sdp_cond_b)
        self._conv(self.sdp_pre, x, H, xs, Fs, B, Tx, Lx, bias_b=cs, bias_b_bs=Fs)
        self._dds(self.sdp_convs, xs, t1, t2, mask, B, Fs, Tx, Lx)
        self._conv(self.sdp_proj, xs, Fs, t1, Fs, B, Tx, Lx, flags=F_MASK_V, mask=mask, mask_bs=Lx)
        xs, t1 = t1, xs                                                         # xs = conditioning of the flows
        if noise_w is None:
            noise_w = torch.randn(B, 2, Tx, dtype=torch.float32, device=dev)
        zw = f(B, 2, Lx)
        zw[:, :, :Tx].copy_(noise_w.to(dev, torch.float32) * float(noise_scale_w))
        c0, c1 = 1, 0                                                           # a Flip precedes every ConvFlow
        for fl in self.sdp_flows:
            _lib.call("ov_expand1_f32", (zw, c0 * Lx), 2 * Lx, fl["pre_w"], fl["pre_b"], xs, hflow, B, Fs, Tx, Lx)
            self._dds(fl["dds"], hflow, t1, t2, mask, B, Fs, Tx, Lx)
            self._conv(fl["proj"], hflow, Fs, hp, 32, B, Tx, Lx, flags=F_MASK_V, mask=mask, mask_bs=Lx)
            _lib.call("ov_rq_spline_inverse_f32", zw, 2 * Lx, c0, c1, hp, 32 * Lx, mask, B, Tx, Lx, SDP_NUM_BINS, Fs,
                      SDP_TAIL_BOUND)
            c0, c1 = c1, c0
        # after the last Flip logical channel 0 is physical channel 0: logw_sdp = EA^-1(z)[0]
        logw = f(B, Lx)
        cum = torch.zeros(B, Lx, dtype=torch.int32, device=dev)
        y_len = torch.zeros(B, dtype=torch.int64, device=dev)
        _lib.call("ov_duration_f32", zw, 2 * Lx, self.ea_m, self.ea_logs, dp_out, 32 * Lx, mask, logw, cum, y_len,
                  B, Tx, Lx, float(sdp_ratio), float(length_scale))
        Ty = int(y_len.max())                                                   # host sync, as in the reference
        # ---- expansion + prior sample + flow (reverse) + generator ---------------------------------------
        core = self.core
        ws = core._workspace(B, Ty)
        Ly, mask_y = ws["Tp"], ws["mask"]
        _lib.call("ov_sequence_mask_f32", y_len, mask_y, B, Ty, Ly)
        if noise_z is None:
            noise_z = torch.randn(B, C, Ty, dtype=torch.float32, device=dev)
        ws["noise"][:, :, :Ty].copy_(noise_z[:, :, :Ty].to(dev, torch.float32))
        m_p, logs_p = f(B, C, Ly), f(B, C, Ly)
        attn = f(B, Ty, Tx) if return_attn else None
        z_p, z = ws["z_p"], ws["z_hat"]
        _lib.call("ov_expand_prior_f32", stats, (stats, C * Lx), 2 * C * Lx, Lx, cum, lengths, y_len, ws["noise"],
                  C * Ly, Ly, z_p, m_p, logs_p, attn, B, C, Tx, Ty, Ly, float(noise_scale))
        cond_flow = [core._wn_cond(cp["wn"], g) for cp in core.couplings]
        core._flow(z_p, z, ws, B, Ty, cond_flow, mask_y, reverse=True)      # z_p -> z, no copy
        cond_d = core._linear(g, core.dec_cond_w, core.dec_cond_b)
        Td = Ty if max_len is None else min(Ty, int(max_len))
        o = core.decode(z, cond_d, ws, T=Td, limits=core.frame_limits(y_len, Td) if skip_padding else None)
        outs = tuple(t[:, :, :Ty].contiguous() for t in (z, z_p, m_p, logs_p))
        self.last_logw = logw[:, :Tx]
        return (o, attn.unsqueeze(1) if attn is not None else None, mask_y[:, :Ty].unsqueeze(1).contiguous(), outs)
