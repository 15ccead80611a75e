"""Host-side logic of the API boundary that needs no GPU: config loader, audio file I/O stand-ins,
watermark bit codec, the alias package, and the fail-loudly behaviour without a ROCm device."""
import json
import struct

import numpy as np
import pytest
import torch

from openvoice_amd import api, audio_io, utils
from openvoice_amd.utils import default_converter_hparams


def _config(tmp_path, version="v2"):
    hps = default_converter_hparams(version)
    cfg = {"data": dict(hps.data.items()), "model": dict(hps.model.items())}
    if version == "v2":
        cfg["_version_"] = "v2"
    path = tmp_path / "config.json"
    path.write_text(json.dumps(cfg))
    return str(path)


def test_hparams_round_trip(tmp_path):
    hps = utils.get_hparams_from_file(_config(tmp_path))
    assert hps.data.sampling_rate == 22050 and hps["data"]["hop_length"] == 256
    assert hps.model.zero_g is True and getattr(hps, "_version_") == "v2"
    kwargs = dict(**hps.model)
    assert kwargs["upsample_rates"] == [8, 8, 2, 2] and "model" in hps and len(hps.data) == 5


def test_no_cpu_path(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU path"):
        api.ToneColorConverter(_config(tmp_path), device="cpu", enable_watermark=False)
    if not torch.cuda.is_available():
        with pytest.raises(AssertionError):          # reference: openvoice/api.py:18-19
            api.ToneColorConverter(_config(tmp_path), device="cuda:0", enable_watermark=False)


def test_engine_rejects_cpu_device(synth_sd):
    from openvoice_amd._lib import OvError
    from openvoice_amd.engine import ConverterEngine
    with pytest.raises(OvError):
        ConverterEngine(synth_sd, utils.CONVERTER_MODEL_CONFIG, 513, "cpu")


def test_v1_tts_model_owns_the_reference_schema(synth_tts_sd):
    """SynthesizerTrn(n_speakers > 0) holds enc_p / sdp / dp / emb_g under the reference's parameter names
    (released base-speaker checkpoints load strictly); without a ROCm device infer() fails loudly."""
    from openvoice_amd._lib import OvError
    from openvoice_amd.models import SynthesizerTrn
    model = SynthesizerTrn(68, 513, n_speakers=10, **utils.CONVERTER_MODEL_CONFIG)
    missing, unexpected = model.load_state_dict(synth_tts_sd, strict=True)
    assert not missing and not unexpected and not hasattr(model, "ref_enc")
    back = model.state_dict()
    assert all(torch.equal(back[k], synth_tts_sd[k]) for k in synth_tts_sd)
    if not torch.cuda.is_available():
        with pytest.raises(OvError):
            model.infer(torch.zeros(1, 3, dtype=torch.long), torch.tensor([3]), sid=torch.tensor([0]))
    conv = SynthesizerTrn(0, 513, n_speakers=0, **utils.CONVERTER_MODEL_CONFIG)
    with pytest.raises(RuntimeError, match="converter variant"):
        conv.infer(torch.zeros(1, 3, dtype=torch.long), torch.tensor([3]))


def test_state_dict_round_trip_strict(synth_sd):
    from openvoice_amd.models import SynthesizerTrn
    model = SynthesizerTrn(0, 513, n_speakers=0, **utils.CONVERTER_MODEL_CONFIG)
    missing, unexpected = model.load_state_dict(synth_sd, strict=True)
    assert not missing and not unexpected
    back = model.state_dict()
    assert set(back) == set(synth_sd)
    assert all(torch.equal(back[k], synth_sd[k]) for k in back)


def test_wav_write_read_round_trip(tmp_path):
    sr = 22050
    t = np.arange(sr // 2, dtype=np.float32) / sr
    x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    path = str(tmp_path / "a.wav")
    audio_io.write(path, x, sr)
    y, sr2 = audio_io.load(path, sr)
    assert sr2 == sr and y.dtype == np.float32 and y.shape == x.shape
    assert np.abs(y - x).max() <= 1.0 / 32768 + 1e-7


def test_wav_reader_formats_and_resample(tmp_path):
    sr = 44100
    t = np.arange(sr // 4) / sr
    x = 0.25 * np.sin(2 * np.pi * 300 * t)
    stereo = np.stack([x, x], 1).astype("<f4")
    body = stereo.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 3, 2, sr, sr * 8, 8, 32) + b"data" + struct.pack("<I", len(body))
    path = tmp_path / "f32_stereo.wav"
    path.write_bytes(hdr + body)
    y, out_sr = audio_io.load(str(path), 22050)
    assert out_sr == 22050 and abs(len(y) - len(x) // 2) <= 1
    ref = 0.25 * np.sin(2 * np.pi * 300 * np.arange(len(y)) / 22050)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3     # polyphase resampler, edges excluded
    with pytest.raises(ValueError):
        bad = tmp_path / "x.mp3"
        bad.write_bytes(b"ID3\x00" * 10)
        audio_io._read_wav(str(bad))


def test_watermark_bit_codec_matches_reference_convention():
    # reference: openvoice/utils.py:46-75 -- MSB-first ASCII, padded to 8 rows of 0b00100000
    bits = api.string_to_bits("@MyShell")
    assert bits.shape == (8, 8) and bits[0].tolist() == [0, 1, 0, 0, 0, 0, 0, 0]
    assert api.bits_to_string(bits) == "@MyShell"
    short = api.string_to_bits("ab")
    assert short[2:].tolist() == [[0, 0, 1, 0, 0, 0, 0, 0]] * 6
    assert api.bits_to_string(short) == "ab" + " " * 6
    assert api.string_to_bits("0123456789").shape == (8, 8)


def test_audio_numpy_concat():
    segs = [np.ones(10, dtype=np.float32), 2 * np.ones((1, 5), dtype=np.float32)]
    out = api.BaseSpeakerTTS.audio_numpy_concat(segs, sr=1000, speed=2.0)
    gap = int(1000 * 0.05 / 2.0)
    assert out.dtype == np.float32 and len(out) == 15 + 2 * gap
    assert out[:10].tolist() == [1.0] * 10 and out[10:10 + gap].sum() == 0 and out[10 + gap] == 2.0


def test_alias_package_exposes_reference_import_paths():
    from openvoice import se_extractor
    from openvoice.api import BaseSpeakerTTS, ToneColorConverter
    assert ToneColorConverter is api.ToneColorConverter and BaseSpeakerTTS is api.BaseSpeakerTTS
    assert callable(se_extractor.get_se)


def test_spectrogram_front_end_matches_oracle(golden_dir):
    import os
    from openvoice_amd.mel_processing import spectrogram_torch
    rec = torch.load(os.path.join(golden_dir, "vc_b2_t17.pt"), weights_only=False)
    spec = spectrogram_torch(rec["wave"], 1024, 22050, 256, 1024, center=False)
    assert (spec - rec["spec"]).abs().max().item() <= 1e-4 * rec["spec"].abs().max().item()


def test_sentence_splitter_and_intersperse():
    pieces = utils.split_sentence("Hello there. This is a longer sentence with more than ten words in it, truly! Ok.",
                                  language_str="EN")
    assert " ".join(pieces).replace("  ", " ") == \
        "Hello there. This is a longer sentence with more than ten words in it, truly! Ok."
    assert all(len(p.split(" ")) >= 3 for p in pieces)
    assert api.intersperse([5, 6, 7], 0) == [0, 5, 0, 6, 0, 7, 0]          # reference: openvoice/commons.py:22-25


def test_sentence_splitter_matches_the_reference_piece_for_piece(golden_dir):
    """tests/golden/split_sentence.json = the UNMODIFIED reference splitter (openvoice/utils.py:78-194) on 54
    (text, language, min_len) cases, generated by oracle/make_split_golden.py: brackets / quotes / language marks in
    user text, comma splitting, count-then-merge, short-piece merging, Chinese punctuation, empty tails."""
    import json
    import os
    with open(os.path.join(golden_dir, "split_sentence.json"), encoding="utf-8") as fh:
        cases = json.load(fh)
    assert len(cases) >= 50
    for c in cases:
        got = utils.split_sentence(c["text"], min_len=c["min_len"], language_str=c["language"])
        assert got == c["pieces"], (c["language"], c["min_len"], c["text"])
    # user text cannot smuggle the language marks BaseSpeakerTTS.tts wraps around every piece
    assert all("[" not in p and "]" not in p for p in utils.split_sentence("say [EN] this [ZH] now.", language_str="EN"))


def test_tts_text_front_end_is_a_hook():
    hps = utils.HParams(symbols=list("_abc"), data=dict(text_cleaners=["x"], add_blank=True))
    with pytest.raises(RuntimeError, match="no text front end"):
        api.BaseSpeakerTTS.get_text("ab", hps, False)
    api.BaseSpeakerTTS.text_to_sequence = staticmethod(lambda text, symbols, cleaners: [symbols.index(c) for c in text])
    try:
        assert api.BaseSpeakerTTS.get_text("abc", hps, False).tolist() == [0, 1, 0, 2, 0, 3, 0]
    finally:
        api.BaseSpeakerTTS.text_to_sequence = None


def test_cleaned_symbol_text_needs_no_front_end():
    """is_symbol=True (already-cleaned text): plain symbol lookup, unknown characters dropped
    (reference: openvoice/text/__init__.py:34-43), blanks interspersed per the config."""
    hps = utils.HParams(symbols=list("_abc"), data=dict(text_cleaners=["x"], add_blank=True))
    assert api.BaseSpeakerTTS.get_text("a?cb", hps, True).tolist() == [0, 1, 0, 3, 0, 2, 0]
    assert utils.cleaned_text_to_sequence("cab!", list("_abc")) == [3, 1, 2]


def test_kaiser_best_resampler_is_band_limited_interpolation():
    """``audio_io.resample_kaiser_best``: the restatement of resampy's published ``kaiser_best`` windowed-sinc resampler
    (what ``librosa.load(path, sr=...)`` applies at the reference's call sites openvoice/api.py:123,144).  Parity with
    resampy itself is UNPINNED (not in this image); checked against what the algorithm must do by construction: a
    sinusoid below the cutoff is reproduced at the new rate (exactly-to-1e-7 where the table step is an integer,
    within the algorithm's own step-truncation gain error elsewhere), one above the new Nyquist is rejected, the output
    length is librosa's ``ceil(n * ratio)``, and the polyphase evaluation equals the per-sample definition."""
    for sr_in, sr_out, bar in [(44100, 22050, 1e-6), (16000, 22050, 1e-6), (48000, 22050, 1e-3), (22050, 16000, 1e-3)]:
        t = np.arange(sr_in // 2) / sr_in
        x = 0.8 * np.sin(2 * np.pi * 997.0 * t + 0.3)
        y = audio_io.resample_kaiser_best(x, sr_in, sr_out)
        assert y.dtype == np.float32 and len(y) == int(np.ceil(len(x) * sr_out / sr_in))
        ref = 0.8 * np.sin(2 * np.pi * 997.0 * np.arange(len(y)) / sr_out + 0.3)
        edge = int(0.05 * sr_out)
        assert np.abs(y[edge:-edge] - ref[edge:-edge]).max() <= bar, (sr_in, sr_out)
        if sr_out < sr_in:                                   # a tone 15 % above the new Nyquist must not alias back
            xs = np.sin(2 * np.pi * 0.575 * sr_out * t)
            assert np.abs(audio_io.resample_kaiser_best(xs, sr_in, sr_out)[edge:-edge]).max() <= 1e-3
    # the definition, sample by sample (resampy.interpn.resample_f), on a short random signal
    rng = np.random.default_rng(0)
    x = rng.standard_normal(700)
    sr_in, sr_out = 48000, 22050
    got = audio_io.resample_kaiser_best(x, sr_in, sr_out).astype(np.float64)
    win = audio_io._sinc_window(**audio_io.KAISER_BEST)
    ratio = sr_out / sr_in
    scale, num_table = min(1.0, ratio), 1 << audio_io.KAISER_BEST["precision"]
    win = win * scale
    delta = np.append(np.diff(win), 0.0)
    step, want = int(scale * num_table), np.zeros(len(got))
    n_res = len(x) * 147 // 320                               # resampy computes int(n * ratio) samples; librosa pads to the ceiling
    assert len(got) == -(-len(x) * 147 // 320) and (got[n_res:] == 0).all()
    for t in range(n_res):
        time = t * 320 / 147                                  # sr_in / sr_out, exactly
        n = int(time)
        for wing, frac in ((0, scale * (time - n)), (1, scale - scale * (time - n))):
            off, eta = int(frac * num_table), frac * num_table - int(frac * num_table)
            i = 0
            while i < (len(win) - off) // step:               # resampy's loop bound
                src = n - i if wing == 0 else n + i + 1
                if 0 <= src < len(x):
                    want[t] += (win[off + i * step] + eta * delta[off + i * step]) * x[src]
                i += 1
    assert np.abs(got - want).max() <= 2e-6                   # float32 output rounding
    assert audio_io.resample(x, 22050, 22050) is x
    with pytest.raises(ValueError):
        audio_io.resample(x, 48000, 22050, "sinc_fastest")


def test_kaiser_best_phase_weights_are_what_both_resampler_forms_apply():
    """``audio_io.kaiser_best_phases``: [P][2 taps] float64 weights shared by the host restatement and the device kernel
    (``ov_polyphase_fir_f32``, tests/test_gpu_resample.py).  The kernel's indexing -- y[t] = h[t % P] . x[(t Q) // P -
    taps + 1 ...] -- restated in numpy equals ``resample_kaiser_best``; at equal rates' simplest ratio (2 : 1) there is
    one phase, symmetric about the sample it sits on."""
    import numpy as np
    from openvoice_amd import audio_io
    h, P, Q, taps = audio_io.kaiser_best_phases(44100, 22050)
    assert (P, Q) == (1, 2) and h.shape == (1, 2 * taps) and h.dtype == np.float64
    assert np.allclose(h[0, :taps - 1][::-1], h[0, taps:2 * taps - 1])      # x[n - k] and x[n + k] weigh the same (k >= 1)
    for sr_in, sr_out in ((48000, 22050), (16000, 22050)):
        h, P, Q, taps = audio_io.kaiser_best_phases(sr_in, sr_out)
        x = np.random.default_rng(sr_in).standard_normal(3000).astype(np.float32)
        want = audio_io.resample_kaiser_best(x, sr_in, sr_out)
        xp = np.concatenate([np.zeros(taps), x.astype(np.float64), np.zeros(taps + Q + 1)])
        n_res = len(x) * P // Q                               # computed samples; the padded tail of `want` is zero
        t = np.arange(n_res)
        n = (t * Q) // P
        got = np.array([h[tt % P] @ xp[nn + 1:nn + 1 + 2 * taps] for tt, nn in zip(t, n)]).astype(np.float32)
        assert np.abs(got - want[:n_res]).max() <= 1e-6 and (want[n_res:] == 0).all()
