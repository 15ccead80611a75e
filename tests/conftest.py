import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_sd():
    """Calibrated synthetic converter weights shared by the whole session."""
    from openvoice_amd.params import synthetic_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG
    return synthetic_state_dict(CONVERTER_MODEL_CONFIG, 513, seed=1234)


@pytest.fixture(scope="session")
def synth_tts_sd():
    """Calibrated synthetic weights of the V1 TTS model (68 symbols, 10 speakers)."""
    from openvoice_amd.params import synthetic_tts_state_dict
    from openvoice_amd.utils import CONVERTER_MODEL_CONFIG
    return synthetic_tts_state_dict(CONVERTER_MODEL_CONFIG, 68, 10, 513, seed=4321)
