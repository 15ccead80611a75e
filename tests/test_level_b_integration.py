"""INTEGRATION.md level B: a maintainer keeps the reference's ``nn.Module`` tree and only swaps the
compute -- ``ConverterEngine(self.model.state_dict(), self.hps.model, ...)`` fed from a module whose convolutions carry
REAL ``torch.nn.utils.weight_norm`` parametrisations (``weight_g`` / ``weight_v``, the ConvTranspose axis included),
as ``openvoice/api.py:35-39`` leaves them after ``load_ckpt``.  The module tree here is rebuilt from the parameter spec
with plain torch layers (the reference itself cannot be imported on the GPU box); loading the calibrated weights into
it ``strict=True`` is the CPU half of the test, the conversion through the patched seam against the oracle the GPU half.
reference: openvoice/models.py:225-270, :307-337, :492-499."""
import pytest
import torch
from torch import nn

from openvoice_amd.params import converter_param_spec
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG

DEV = "cuda:0"


def reference_shaped_module(cfg=CONVERTER_MODEL_CONFIG, spec_channels=513):
    """An ``nn.Module`` with the reference converter's parameter names and shapes, weight-normed where the reference is."""
    spec = converter_param_spec(spec_channels, **cfg)
    leaves = {}
    for name, shape in spec.items():
        path, leaf = name.rsplit(".", 1)
        leaves.setdefault(path, {})[leaf] = tuple(shape)
    root = nn.Module()
    for path, params in leaves.items():
        if "weight_ih_l0" in params:                                   # ref_enc.gru (models.py:327-329)
            hid = params["weight_hh_l0"][1]
            mod = nn.GRU(params["weight_ih_l0"][1], hid, batch_first=True)
        else:
            w = params.get("weight_v", params.get("weight"))
            bias = "bias" in params
            if len(w) == 1:
                mod = nn.LayerNorm(w[0])
            elif len(w) == 2:
                mod = nn.Linear(w[1], w[0], bias=bias)
            elif len(w) == 4:
                mod = nn.Conv2d(w[1], w[0], (w[2], w[3]), bias=bias)
            elif path.startswith("dec.ups."):
                mod = nn.ConvTranspose1d(w[0], w[1], w[2], bias=bias)  # weight is [C_in, C_out, K] (models.py:244-256)
            else:
                mod = nn.Conv1d(w[1], w[0], w[2], bias=bias)
            if "weight_v" in params:
                mod = torch.nn.utils.weight_norm(mod)                  # the parametrisation the reference applies
                assert tuple(mod.weight_g.shape) == params["weight_g"], (path, mod.weight_g.shape, params["weight_g"])
        parent = root
        parts = path.split(".")
        for part in parts[:-1]:
            if not hasattr(parent, part):
                parent.add_module(part, nn.Module())
            parent = getattr(parent, part)
        parent.add_module(parts[-1], mod)
    return root


def test_reference_shaped_module_loads_the_calibrated_weights_strictly(synth_sd):
    """Key names, shapes and the weight-norm ``g`` axes of real torch modules are exactly the state-dict the engine and
    the oracle consume (CPU)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = reference_shaped_module()
    res = mod.load_state_dict(synth_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    sd = mod.state_dict()
    assert list(sd) == list(synth_sd) or set(sd) == set(synth_sd)
    # a ConvTranspose's g is per INPUT channel (dim 0 of [C_in, C_out, K]), as torch's weight_norm lays it out
    assert sd["dec.ups.0.weight_g"].shape == (512, 1, 1) and sd["dec.ups.0.weight_v"].shape == (512, 256, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("zero_g", [True, False])
def test_level_b_patch_matches_the_oracle(synth_sd, zero_g):
    """The maintainer-side patch of INTEGRATION.md section 3, verbatim: engine from ``model.state_dict()``, the seam's
    ``voice_conversion`` / ``ref_enc`` replaced, results against the oracle at the fp32 bars."""
    import warnings
    from types import SimpleNamespace

    from openvoice_amd.engine import ConverterEngine
    from openvoice_amd.utils import default_converter_hparams
    from oracle import vc_oracle
    hps = default_converter_hparams("v2")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = reference_shaped_module(dict(hps.model.items()), hps.data.filter_length // 2 + 1)
    model.load_state_dict(synth_sd, strict=True)
    model.zero_g = zero_g
    self = SimpleNamespace(model=model.to(DEV), hps=hps, device=DEV)
    # ---- the patch -----------------------------------------------------------------------
    eng = ConverterEngine(self.model.state_dict(), self.hps.model, self.hps.data.filter_length // 2 + 1, self.device,
                          zero_g=getattr(self.model, 'zero_g', False))
    self.model.voice_conversion = eng.voice_conversion
    self.model.ref_enc.forward = lambda x, mask=None: eng.reference_encoder(x)
    # -------------------------------------------------------------------------------------------------------------------
    B, T = 2, 96
    gen = torch.Generator().manual_seed(11)
    spec = torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]
    g_src, g_tgt = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
    noise = torch.randn(B, 192, T, generator=gen)
    lengths = torch.tensor([T, T - 9])
    o_hat, y_mask, (z, z_p, z_hat) = self.model.voice_conversion(spec.to(DEV), lengths.to(DEV), g_src.to(DEV), g_tgt.to(DEV),
                                                                 tau=0.3, noise=noise.to(DEV))
    with torch.no_grad():
        o_ref, mask_ref, (z_r, zp_r, zh_r) = vc_oracle.voice_conversion(synth_sd, CONVERTER_MODEL_CONFIG, spec, lengths, g_src,
                                                                        g_tgt, 0.3, noise, zero_g=zero_g)
        se_ref = vc_oracle.reference_encoder(synth_sd, spec.transpose(1, 2))
    assert torch.equal(y_mask.cpu(), mask_ref) and o_hat.shape == o_ref.shape and o_hat.dtype == torch.float32
    assert max((z.cpu() - z_r).abs().max(), (z_p.cpu() - zp_r).abs().max(), (z_hat.cpu() - zh_r).abs().max()) <= 2e-4
    assert (o_hat.cpu() - o_ref).abs().max() <= 1e-3
    se = self.model.ref_enc.forward(spec.transpose(1, 2).contiguous().to(DEV))
    assert se.shape == se_ref.shape and (se.cpu() - se_ref).abs().max() <= 1e-4
