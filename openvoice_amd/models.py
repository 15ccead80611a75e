"""``SynthesizerTrn`` -- the model seam of the reference API, re-hosted on the HIP engine.

Keeps the constructor signature, parameter names (so released ``checkpoint.pth`` files load with
``load_state_dict``) and the inference entry points of the reference class
(reference: openvoice/models.py:399-499) but contains no eager PyTorch compute: ``voice_conversion``
and ``ref_enc`` run on ``ConverterEngine`` and ``infer`` on ``TtsEngine`` (hand-written gfx950
kernels).  ``n_speakers == 0`` builds the converter variant (``enc_q``, ``flow``, ``dec``, ``ref_enc``),
``n_speakers > 0`` the V1 base-speaker TTS variant (adds ``enc_p``, ``sdp``, ``dp``, ``emb_g``).
"""
import weakref

import torch
from torch import nn

from .engine import ConverterEngine, validate_config
from .params import converter_param_spec, tts_full_param_spec


def _attach(root, dotted, tensor):
    parts = dotted.split(".")
    mod = root
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, nn.Module())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _ReferenceEncoderHandle(nn.Module):
    """``model.ref_enc`` stays a parameter-owning, callable submodule as in the reference
    (openvoice/models.py:301-364, called at openvoice/api.py:131); its forward runs on the engine."""

    def __init__(self, root):
        super().__init__()
        object.__setattr__(self, "_root", weakref.ref(root))

    def forward(self, inputs, mask=None):
        return self._root().engine().reference_encoder(inputs)


class SynthesizerTrn(nn.Module):
    def __init__(self, n_vocab, spec_channels, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, n_speakers=256,
                 gin_channels=256, zero_g=False, **kwargs):
        super().__init__()
        self.n_speakers = n_speakers
        self.n_vocab = n_vocab
        self.zero_g = zero_g
        self.spec_channels = spec_channels
        self.model_cfg = dict(
            inter_channels=inter_channels, hidden_channels=hidden_channels, resblock=resblock,
            resblock_kernel_sizes=list(resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
            upsample_rates=list(upsample_rates), upsample_initial_channel=upsample_initial_channel,
            upsample_kernel_sizes=list(upsample_kernel_sizes), gin_channels=gin_channels,
            filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers, kernel_size=kernel_size)
        validate_config(self.model_cfg)     # an unsupported hyper-parameter is named here, not as a state-dict key error
        if n_speakers == 0:     # converter variant: ref_enc, no text side (reference: models.py:450-452)
            self.ref_enc = _ReferenceEncoderHandle(self)
            spec = converter_param_spec(spec_channels, **self.model_cfg)
        else:                   # V1 base-speaker TTS: enc_p, sdp, dp, emb_g (reference: models.py:453-464)
            spec = tts_full_param_spec(n_vocab, n_speakers, spec_channels, **self.model_cfg)
        gen = torch.Generator().manual_seed(0)
        for name, shape in spec.items():
            if name.endswith(("weight_g", ".gamma")) or name == "ref_enc.layernorm.weight":
                init = torch.ones(shape)
            elif len(shape) == 1:
                init = torch.zeros(shape)
            else:
                init = 0.01 * torch.randn(shape, generator=gen)   # reference: openvoice/commons.py:6-9
            _attach(self, name, init)
        self._engine = None
        self._engine_key = None

    # The engine snapshots (folds + packs) the parameters; rebuild it when they change or move.
    def _apply(self, fn, *args, **kwargs):
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._engine = None
        return super().load_state_dict(*args, **kwargs)

    def engine(self):
        dev = next(self.parameters()).device
        key = (str(dev), bool(self.zero_g))
        if self._engine is None or self._engine_key != key:
            if self.n_speakers == 0:
                self._engine = ConverterEngine(self.state_dict(), self.model_cfg, self.spec_channels, dev,
                                               zero_g=self.zero_g)
            else:
                from .tts_engine import TtsEngine
                self._engine = TtsEngine(self.state_dict(), self.model_cfg, self.spec_channels, dev,
                                         self.n_vocab, self.n_speakers)
            self._engine_key = key
        return self._engine

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, tau=1.0, noise=None, graph=False, skip_padding=False):
        """reference: openvoice/models.py:492-499; ``noise`` is the explicit form of the
        reference's ``randn_like`` draw (optional).  ``graph=True`` replays the launch sequence of this
        (B, T, tau) shape from a captured HIP graph (``engine.GraphedConversion``); the returned tensors are
        then static buffers, valid until the next graphed call of the same shape.  ``skip_padding=True`` (ragged
        batches): the generator computes only ``length + 16`` frames per utterance -- valid samples bit-identical,
        the padded tail of ``o_hat`` zero (``ConverterEngine.voice_conversion``)."""
        eng = self.engine()
        if self.n_speakers != 0:
            eng = eng.core      # a TTS checkpoint also carries enc_q / flow / dec
        if graph:
            g = eng.graphed(y.shape[0], y.shape[2], tau, sid_src.shape[0], sid_tgt.shape[0], skip_padding=skip_padding)
            return g(y, y_lengths, sid_src, sid_tgt, noise=noise)
        return eng.voice_conversion(y, y_lengths, sid_src, sid_tgt, tau=tau, noise=noise, skip_padding=skip_padding)

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1., sdp_ratio=0.2,
              max_len=None, noise_w=None, noise_z=None, skip_padding=False):
        """reference: openvoice/models.py:467-490; ``noise_w`` / ``noise_z`` are the explicit forms of the
        reference's two RNG draws (optional).  Returns ``(o, attn, y_mask, (z, z_p, m_p, logs_p))``."""
        if self.n_speakers == 0:
            raise RuntimeError("infer() needs the TTS model (n_speakers > 0); this is the converter variant")
        return self.engine().infer(x, x_lengths, sid, noise_scale=noise_scale, length_scale=length_scale,
                                   noise_scale_w=noise_scale_w, sdp_ratio=sdp_ratio, max_len=max_len,
                                   noise_w=noise_w, noise_z=noise_z, skip_padding=skip_padding)
