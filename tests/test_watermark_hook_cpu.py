"""The optional watermark hook of ``ToneColorConverter`` (reference: openvoice/api.py:162-201; the model itself is the
third-party wavmark and is never shipped) exercised with a stub model: window indexing (group n in the 16 000 samples
from 32 000 n), the 8-character / 64-bit message codec round trip, and the "audio too short" branches."""
import numpy as np
import torch

from openvoice_amd import api


class StubMark:
    """encode adds 1e-3 * mean(bits) to the window and remembers the bits; decode returns them in order."""

    def __init__(self):
        self.sent, self.windows = [], []

    def encode(self, signal, msg):
        assert signal.shape == (1, 16000) and msg.shape == (1, 32)
        self.sent.append(msg.clone())
        self.windows.append(signal.clone())
        return signal + 1e-3 * msg.mean()

    def decode(self, signal):
        assert signal.shape == (1, 16000)
        return self.sent.pop(0)


def _converter(model):
    conv = object.__new__(api.ToneColorConverter)      # the hook needs neither weights nor a GPU
    conv.device = "cpu"
    conv.watermark_model = model
    return conv


def test_message_round_trip_and_window_indexing():
    model = StubMark()
    conv = _converter(model)
    audio = np.linspace(-1.0, 1.0, 80000, dtype=np.float32)
    before = audio.copy()
    out = conv.add_watermark(audio, "openAI!?")
    assert out is audio and len(model.sent) == 2                 # 8 characters = 64 bits = two 32-bit groups
    bits = api.string_to_bits("openAI!?").reshape(-1)
    for n in range(2):
        lo = 32000 * n
        assert torch.equal(model.windows[n][0], torch.from_numpy(before[lo:lo + 16000]))
        want = before[lo:lo + 16000] + np.float32(1e-3 * bits[32 * n:32 * n + 32].mean())
        assert np.allclose(audio[lo:lo + 16000], want, atol=1e-7)
    untouched = np.ones(80000, bool)
    untouched[0:16000] = untouched[32000:48000] = False
    assert np.array_equal(audio[untouched], before[untouched])
    assert conv.detect_watermark(audio, 2) == "openAI!?"


def test_short_messages_are_space_padded_and_long_ones_cut():
    assert api.bits_to_string(api.string_to_bits("hi")) == "hi      "
    assert api.bits_to_string(api.string_to_bits("0123456789")) == "01234567"


def test_audio_too_short(capsys):
    model = StubMark()
    conv = _converter(model)
    audio = np.zeros(40000, dtype=np.float32)                    # room for window 0 only (window 1 needs 48 000 samples)
    conv.add_watermark(audio, "default")
    assert len(model.sent) == 1 and "Audio too short, fail to add watermark" in capsys.readouterr().out
    assert conv.detect_watermark(audio, 2) == "Fail"
    assert "Audio too short, fail to detect watermark" in capsys.readouterr().out
    conv.watermark_model = None                                  # hook disabled: the waveform passes through
    assert conv.add_watermark(audio, "x") is audio
