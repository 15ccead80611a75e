"""The device resampler of the audio boundary (``ov_polyphase_fir_f32`` + ``audio_io.resample_on_device``; reference:
openvoice/api.py:123,144 ``librosa.load(path, sr=...)`` = resampy ``kaiser_best``) against its host restatement
(``audio_io.resample_kaiser_best``: the same float64 phase weights, BLAS summation order), through the C ABI.
Tolerance: 1e-6 absolute on unit-scale signals (both sides accumulate in float64 and round once to float32; the summation
orders differ)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib, audio_io  # noqa: E402

DEV = "cuda:0"
PAIRS = [(44100, 22050), (48000, 22050), (16000, 22050), (24000, 22050), (8000, 22050), (22050, 16000), (32000, 22050)]


@pytest.mark.parametrize("sr_in,sr_out", PAIRS)
@pytest.mark.parametrize("n", [1, 7, 1000, 50001])
def test_device_resampler_matches_the_host_restatement(sr_in, sr_out, n):
    x = np.random.default_rng(n + sr_in).standard_normal(n).astype(np.float32)
    want = audio_io.resample_kaiser_best(x, sr_in, sr_out)
    got = audio_io.resample_on_device(torch.from_numpy(x).to(DEV), sr_in, sr_out).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max()))


def test_ten_seconds_of_a_band_limited_signal_keep_their_closed_form():
    """44.1 -> 22.05 kHz on the device: sinusoids below the new Nyquist come out as the same sinusoids sampled at the new
    rate (the property the host restatement is tested on, tests/test_api_cpu.py), one above it is rejected."""
    sr_in, sr_out, secs = 44100, 22050, 10
    t_in, t_out = np.arange(secs * sr_in) / sr_in, np.arange(secs * sr_out) / sr_out
    for f, kept in ((440.0, True), (5000.0, True), (9000.0, True), (15000.0, False)):
        x = np.sin(2 * np.pi * f * t_in).astype(np.float32)
        y = audio_io.resample_on_device(torch.from_numpy(x).to(DEV), sr_in, sr_out).cpu().numpy()
        inner = slice(2000, -2000)                                        # away from the edges (zero extension)
        if kept:
            assert np.abs(y[inner] - np.sin(2 * np.pi * f * t_out)[inner]).max() <= 2e-4, f
        else:
            assert np.abs(y[inner]).max() <= 1e-3, f


def test_load_to_device_equals_load(tmp_path):
    """``load_to_device`` = ``load`` with the rate conversion moved to the GPU: a 44.1 kHz stereo WAV (mixed down, then
    resampled) and a file already at the model rate (no resampling: identical bits)."""
    sr = 44100
    t = np.arange(int(1.7 * sr)) / sr
    stereo = np.stack([0.4 * np.sin(2 * np.pi * 330 * t), 0.3 * np.sin(2 * np.pi * 1234 * t + 0.5)], axis=1).astype(np.float32)
    audio_io.write(str(tmp_path / "a.wav"), stereo, sr)
    want, _ = audio_io.load(str(tmp_path / "a.wav"), 22050)
    got = audio_io.load_to_device(str(tmp_path / "a.wav"), 22050, DEV)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == (len(want),)
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-6
    audio_io.write(str(tmp_path / "b.wav"), stereo[:, 0], 22050)
    assert np.array_equal(audio_io.load_to_device(str(tmp_path / "b.wav"), 22050, DEV).cpu().numpy(),
                          audio_io.load(str(tmp_path / "b.wav"), 22050)[0])


def test_argument_checks():
    x = torch.zeros(16, device=DEV)
    h = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    y = torch.zeros(8, device=DEV)
    with pytest.raises(_lib.OvError):
        _lib.call("ov_polyphase_fir_f32", x, h, y, 16, 8, 0, 2, 2)        # P = 0
    with pytest.raises(_lib.OvError):
        _lib.call("ov_polyphase_fir_f32", x, None, y, 16, 8, 1, 2, 2)     # no weights
    _lib.call("ov_polyphase_fir_f32", x, h, y, 16, 8, 1, 2, 2)
    torch.cuda.synchronize()
