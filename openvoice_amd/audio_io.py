"""Audio file boundary of the converter API: decode -> mono float32 at the model rate, and WAV write.

The reference delegates both to third-party packages that are not part of its tree:
``librosa.load(path, sr=hps.data.sampling_rate)`` (reference: openvoice/api.py:123,144) and
``soundfile.write`` (openvoice/api.py:98,160).  When those packages are importable they are used,
so behaviour is then the reference's by construction.  When they are absent (this image), a small
RIFF/WAVE reader + polyphase resampler stands in: PCM 8/16/24/32-bit and IEEE float32/64 WAV,
channel mean for mono, ``scipy.signal.resample_poly`` for rate conversion.  Parity of this
stand-in with librosa's ``soxr_hq``/``kaiser_best`` resampler is NOT pinned (SURVEY.md section 8c
item (i)): the parity boundary of this repo starts at the float32 waveform at the model rate.
Compressed formats (mp3 ...) need librosa/audioread and raise a clear error without them.
"""
import struct
from math import gcd

import numpy as np


def _read_wav(path):
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (install librosa to decode other formats)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if code == 0xFFFE and len(body) >= 26:       # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                code = struct.unpack("<H", body[24:26])[0]
            fmt = (code, channels, rate, bits)
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    code, channels, rate, bits = fmt
    if code == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            raw = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            val = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
            val = np.where(val >= 1 << 23, val - (1 << 24), val)
            x = val.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / float(1 << 31)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif code == 3:
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format code {code}")
    x = x[:len(x) // channels * channels].reshape(-1, channels)
    return x, rate


def resample(x, sr_in, sr_out):
    if sr_in == sr_out:
        return x
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(x.astype(np.float64), int(sr_out) // g, int(sr_in) // g).astype(np.float32)


def load(path, sr):
    """``(mono float32 waveform at ``sr``, sr)`` -- the contract of ``librosa.load(path, sr=sr)``."""
    try:
        import librosa  # noqa: F401  (third-party; used when present so decoding equals the reference's)
        return librosa.load(path, sr=sr)
    except ImportError:
        pass
    x, rate = _read_wav(path)
    mono = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    return np.ascontiguousarray(resample(mono, rate, sr), dtype=np.float32), sr


def write(path, audio, sr):
    """``soundfile.write(path, audio, sr)`` for a mono float waveform: 16-bit PCM WAV (soundfile's
    default subtype for .wav)."""
    try:
        import soundfile
        return soundfile.write(path, audio, sr)
    except ImportError:
        pass
    audio = np.asarray(audio, dtype=np.float32).reshape(-1)
    pcm = np.clip(np.rint(audio * 32768.0), -32768, 32767).astype("<i2").tobytes()
    header = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, int(sr), int(sr) * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    with open(path, "wb") as fh:
        fh.write(header + pcm)
