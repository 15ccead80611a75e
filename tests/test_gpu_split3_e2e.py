"""The whole conversion with the generator's MRF stages on the split-precision kernels
(``ConverterEngine.use_split_bf16x3()``, csrc/conv1d_split3.h; VERDICT r04 item 1: "show every fp32 parity test passing at
the fp32 bars -- latents 2e-4, o_hat <= 1e-4 measured, not merely <= 1e-3").  The fp32 end-to-end parity tests of
tests/test_gpu_e2e.py are re-run here, unchanged, with every engine built in this module switched to the split path and
the waveform bar tightened from 1e-3 to 1e-4: the reference goldens (incl. the gain-4 stress case), the oracle shape list
(T = 1 ... 1000, ragged, per-item embeddings), the batch-32 x 861-frame benchmark shape, and the non-released
configuration.  reference: openvoice/models.py:272-291, modules.py:296-309."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_gpu_e2e as fp32_tests  # noqa: E402
from openvoice_amd.engine import ConverterEngine  # noqa: E402

SPLIT_O_HAT_TOL = 1e-4


@pytest.fixture(autouse=True)
def split_precision_engines(monkeypatch):
    built = []
    original = ConverterEngine.__init__

    def init(self, *args, **kwargs):
        original(self, *args, **kwargs)
        self.use_split_bf16x3(True)
        built.append(self)

    monkeypatch.setattr(ConverterEngine, "__init__", init)
    monkeypatch.setattr(fp32_tests, "O_HAT_TOL", SPLIT_O_HAT_TOL)
    yield built
    assert built, "the test built no engine"
    for eng in built:       # the split path really ran: at least one stage has split-precision instances
        assert eng._split3_on and any(st is not None for st in eng.split_resblocks)


test_voice_conversion_matches_reference_golden = fp32_tests.test_voice_conversion_matches_reference_golden
test_voice_conversion_matches_oracle = fp32_tests.test_voice_conversion_matches_oracle
test_non_released_config_matches_oracle = fp32_tests.test_non_released_config_matches_oracle


def test_benchmark_shape_items_match_the_oracle_at_the_fp32_bar(synth_sd):
    """B = 32 x T = 861 (BASELINE.json configs[1]): two items of the batch against the oracle, split path vs fp32 path on
    the same inputs; reports how far the two GPU paths are from each other and from the oracle."""
    from openvoice_amd.models import SynthesizerTrn
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG
    from oracle import vc_oracle
    B, T, dev = 32, 861, "cuda:0"
    gen = torch.Generator().manual_seed(4321)
    spec = torch.rand(B, 513, T, generator=gen).abs() * torch.linspace(3, 0.05, 513)[None, :, None]
    g_src, g_tgt = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
    noise = torch.randn(B, 192, T, generator=gen)
    lengths = torch.full((B,), T, dtype=torch.long)
    model = SynthesizerTrn(0, 513, n_speakers=0, zero_g=True, **CONVERTER_MODEL_CONFIG)
    model.load_state_dict(synth_sd, strict=True)
    model = model.to(dev).eval()
    eng = model.engine()
    run = lambda: model.voice_conversion(spec.to(dev), lengths.to(dev), g_src.to(dev), g_tgt.to(dev), tau=0.3,
                                         noise=noise.to(dev))[0].cpu()
    o_split = run()
    eng.use_split_bf16x3(False)
    o_fp32 = run()
    eng.use_split_bf16x3(True, products=3)
    o_three = run()
    eng.use_split_bf16x3(True)
    from openvoice_amd.hostinfo import usable_cpus
    torch.set_num_threads(usable_cpus(32))
    for item in (3, 20):
        with torch.no_grad():
            want = vc_oracle.voice_conversion(synth_sd, CONVERTER_MODEL_CONFIG, spec[item:item + 1], lengths[item:item + 1],
                                              g_src, g_tgt, 0.3, noise[item:item + 1], zero_g=True)[0][0]
        e_split = (o_split[item] - want).abs().max().item()
        e_fp32 = (o_fp32[item] - want).abs().max().item()
        e_three = (o_three[item] - want).abs().max().item()
        print(f"item {item}: vs oracle split(6) {e_split:.2e}, fp32 kernels {e_fp32:.2e}, split(3 products) {e_three:.2e}; "
              f"split vs fp32 kernels {(o_split[item] - o_fp32[item]).abs().max().item():.2e}; |o|max {want.abs().max().item():.3f}")
        assert e_split <= SPLIT_O_HAT_TOL and e_fp32 <= SPLIT_O_HAT_TOL
        assert e_three <= 1e-3             # the 16-bit-operand mode is held to BASELINE.json's 1e-3, not to the fp32 bar


import test_gpu_fuzz as fuzz_tests  # noqa: E402


@pytest.mark.parametrize("case", fuzz_tests._cases(24, 7)[::3],
                         ids=lambda c: f"B{c['B']}_T{c['T']}_{'skip' if c['skip'] else 'full'}")
def test_random_shapes_with_the_split_path(synth_sd, case):
    """Every third case of tests/test_gpu_fuzz.py (random batch / frames / ragged lengths / zero_g / tau, half of them with the
    length-aware work lists) through engines built with the split path on.  Under ``skip_padding`` the split stages compute
    the whole tensors (a superset), the fp32 stage and conv_post still honour the limits: the valid samples match the oracle
    and the padded tail is exact silence."""
    fuzz_tests._models.clear()                       # (models cached by the fp32 fuzz run were built without the split path)
    try:
        fuzz_tests.test_random_shape_matches_oracle(synth_sd, case)
    finally:
        fuzz_tests._models.clear()
