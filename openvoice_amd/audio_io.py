"""Audio file boundary of the converter API: decode -> mono float32 at the model rate, and WAV write.

The reference delegates both to third-party packages that are not part of its tree:
``librosa.load(path, sr=hps.data.sampling_rate)`` (reference: openvoice/api.py:123,144) and
``soundfile.write`` (openvoice/api.py:98,160).  When those packages are importable they are used,
so behaviour is then the reference's by construction.  When they are absent (this image), a small
RIFF/WAVE reader + a restatement of resampy's published ``kaiser_best`` windowed-sinc resampler (what
librosa 0.9.1's ``load(sr=...)`` applies) stand in: PCM 8/16/24/32-bit and IEEE float32/64 WAV, channel
mean for mono (``librosa.to_mono``), ``resample_kaiser_best`` for rate conversion.  Parity of the resampler
with resampy's own output is NOT pinned (neither package is in this image): it
follows the published algorithm and filter constants and is tested against closed-form band-limited
interpolation; the parity boundary of this repo starts at the float32 waveform at the model rate.
MP3 (MPEG-1 / 2 / 2.5 Layer III; resources/*.mp3 are MPEG-1) is decoded by ``openvoice_amd.mp3``, a decoder written from the
standard's decoding process and PINNED against FFmpeg's output (Chromium's build of it: oracle/make_mp3_golden.py,
tests/test_mp3_cpu.py: identical sample counts, max-abs 5e-5 = 1.5 LSB of 16-bit PCM on the reference's four files).
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


# resampy's published ``kaiser_best`` filter (librosa 0.9.1's default ``res_type``, reference call sites
# openvoice/api.py:123,144 through ``librosa.load(path, sr=...)``): a Kaiser-windowed sinc with 64 zero crossings,
# 2^9 table samples per crossing, these two constants
KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)
_kaiser_best_window = None


def _sinc_window(num_zeros, precision, rolloff, beta):
    """Right half of the interpolation filter, ``num_zeros * 2^precision + 1`` samples (resampy.filters.sinc_window):
    ``rolloff * sinc(rolloff * t)`` for t in [0, num_zeros], tapered by the right half of a Kaiser(beta) window."""
    n = (1 << precision) * num_zeros
    t = np.linspace(0.0, num_zeros, num=n + 1, endpoint=True)
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * (rolloff * np.sinc(rolloff * t))


def kaiser_best_phases(sr_in, sr_out):
    """``(h [P][2 taps] float64, P, Q, taps)`` -- the interpolation weights of resampy's ``kaiser_best`` resampler at each
    of the P fractional positions an output sample can take (P / Q = sr_out / sr_in in lowest terms): output t sits at
    input time ``t * Q / P`` = n + rem / P with n = (t * Q) // P, and
    ``y[t] = sum_j h[t % P][j] * x[n - taps + 1 + j]``.  The filter table is read at steps of ``int(scale * 2^precision)``
    entries (``scale = min(1, sr_out / sr_in)``: the cutoff follows the lower rate) from offset
    ``scale * frac * 2^precision`` with linear interpolation between entries -- left wing over x[n], x[n-1], ..., right
    wing (offset from ``scale * (1 - frac)``) over x[n+1], ... -- and the table is scaled by ``scale`` when downsampling
    (resampy.interpn.resample_f).  Shared by the host restatement below and the device kernel
    (``ov_polyphase_fir_f32``)."""
    global _kaiser_best_window
    if _kaiser_best_window is None:
        _kaiser_best_window = _sinc_window(**KAISER_BEST)
    g = gcd(int(sr_in), int(sr_out))
    P, Q = int(sr_out) // g, int(sr_in) // g                  # input time of output t = t * Q / P
    ratio = P / Q
    scale = min(1.0, ratio)
    num_table = 1 << KAISER_BEST["precision"]
    win = _kaiser_best_window * scale if ratio < 1.0 else _kaiser_best_window
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    nwin = win.shape[0]
    index_step = int(scale * num_table)
    taps = (nwin - 1) // index_step + 1                       # an upper bound of either wing's length
    k = np.arange(taps)
    h = np.zeros((P, 2 * taps))                               # row r: weights of x[n - taps + 1 ... n + taps]
    for r in range(P):
        rem = (r * Q) % P
        for wing, f in ((0, scale * rem / P), (1, scale - scale * rem / P)):
            index_frac = f * num_table
            offset = int(index_frac)
            eta = index_frac - offset
            idx = offset + k * index_step
            ok = k < (nwin - offset) // index_step              # resampy's loop bound (one tap short of idx < nwin at most)
            w = np.where(ok, win[np.minimum(idx, nwin - 1)] + eta * delta[np.minimum(idx, nwin - 1)], 0.0)
            if wing == 0:
                h[r, taps - 1 - k] = w                        # x[n - k]
            else:
                h[r, taps + k] = w                            # x[n + k + 1]
    return h, P, Q, taps


def resample_kaiser_best(x, sr_in, sr_out):
    """Band-limited sinc interpolation, a restatement of resampy's published algorithm (Smith's "Digital Audio
    Resampling", resampy.interpn.resample_f) with its ``kaiser_best`` filter (weights: ``kaiser_best_phases``).
    Evaluated phase by phase: with integer rates the fractional position takes ``sr_out / gcd`` values, each one fixed
    set of weights applied to a strided window view of x (one matrix-vector product per phase) instead of resampy's
    per-sample loops; the weights are the algorithm's, the summation order is BLAS's.  resampy computes
    ``int(len * ratio)`` samples and ``librosa.resample(fix=True)`` zero-pads them to ``ceil(len * ratio)``: the same
    here (the one extra sample of a non-integer product is 0, not interpolated).
    PARITY UNPINNED: neither resampy nor librosa is in this image, so agreement with their output is by construction of
    the published algorithm only (tests: closed-form band-limited interpolation of sinusoids, stop-band rejection)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if int(sr_in) == int(sr_out) or x.size == 0:
        return x.astype(np.float32)
    h, P, Q, taps = kaiser_best_phases(sr_in, sr_out)
    n_orig = x.shape[0]
    n_out = n_orig * P // Q                                   # what resampy computes; padded to the ceiling below
    xpad = np.concatenate([np.zeros(taps), x, np.zeros(taps + Q + 1)])   # x[i] = xpad[i + taps]
    y = np.zeros(n_out, dtype=np.float64)
    for r in range(min(P, n_out)):
        n0 = (r * Q) // P                                     # outputs t = r + P m sit at n0 + m Q + rem / P
        m = (n_out - 1 - r) // P + 1
        first = n0 + 1                                        # xpad index of x[n0 - taps + 1]
        view = np.lib.stride_tricks.as_strided(xpad[first:], shape=(m, 2 * taps), strides=(Q * xpad.strides[0], xpad.strides[0]),
                                               writeable=False)
        y[r::P] = view @ h[r]
    n_fix = -(-n_orig * P // Q)
    return np.concatenate([y, np.zeros(n_fix - n_out)]).astype(np.float32)


_device_phases = {}


def resample_on_device(x, sr_in, sr_out):
    """The same resampler on the GPU (``ov_polyphase_fir_f32``): ``x`` a float32 DEVICE tensor [N] -> float32 device tensor
    [ceil(N * sr_out / sr_in)]; the phase weights are computed once per (rate pair, device) on the host and kept
    resident."""
    import torch
    from . import _lib
    if int(sr_in) == int(sr_out) or x.numel() == 0:
        return x
    key = (int(sr_in), int(sr_out), str(x.device))
    if key not in _device_phases:
        h, P, Q, taps = kaiser_best_phases(sr_in, sr_out)
        _device_phases[key] = (torch.from_numpy(h).to(x.device), P, Q, taps)
    h, P, Q, taps = _device_phases[key]
    x = x.to(torch.float32).contiguous()
    n_in = int(x.numel())
    n_res, n_out = n_in * P // Q, -(-n_in * P // Q)           # resampy's sample count, librosa's fixed length
    y = torch.zeros(n_out, dtype=torch.float32, device=x.device) if n_out > n_res else torch.empty(
        n_out, dtype=torch.float32, device=x.device)
    if n_res > 0:
        _lib.call("ov_polyphase_fir_f32", x, h, y, n_in, n_res, P, Q, taps)
    return y


def resample(x, sr_in, sr_out, res_type="kaiser_best"):
    """Rate conversion of a mono waveform.  ``kaiser_best`` (default): the restatement above of what
    ``librosa.load(path, sr=...)`` applies in the reference (librosa 0.9.1 -> resampy ``kaiser_best``); ``polyphase``:
    ``scipy.signal.resample_poly`` (round 1-4's stand-in, kept for comparison)."""
    if sr_in == sr_out:
        return x
    if res_type == "kaiser_best":
        return resample_kaiser_best(x, sr_in, sr_out)
    if res_type != "polyphase":
        raise ValueError(f"unknown res_type {res_type!r}")
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(x.astype(np.float64), int(sr_out) // g, int(sr_in) // g).astype(np.float32)


def read_native(path):
    """``(float32 [samples, channels], file rate)`` of a WAV or MPEG-1 Layer III file, no resampling."""
    with open(path, "rb") as fh:
        head = fh.read(12)
    if head[:4] == b"RIFF" and head[8:12] == b"WAVE":
        return _read_wav(path)
    from . import mp3
    with open(path, "rb") as fh:
        pcm, rate = mp3.decode(fh.read())
    return pcm.T, rate


def load(path, sr):
    """``(mono float32 waveform at ``sr``, sr)`` -- the contract of ``librosa.load(path, sr=sr)``; ``sr=None`` keeps the
    file's own rate."""
    try:
        import librosa  # noqa: F401  (third-party; used when present so decoding equals the reference's)
        return librosa.load(path, sr=sr)
    except ImportError:
        pass
    # RIFF/WAVE natively; anything else is tried as MPEG-1 Layer III (resources/*.mp3 of the reference, BASELINE.json
    # configs[0]) with the from-scratch decoder of openvoice_amd/mp3.py, pinned against FFmpeg's output
    x, rate = read_native(path)
    mono = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    if sr is None:
        return np.ascontiguousarray(mono, dtype=np.float32), rate
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


def load_to_device(path, sr, device):
    """``librosa.load(path, sr=sr)`` with the waveform ending up on ``device``: the file is decoded on the host (WAV /
    MP3), mixed down, and -- when its rate differs -- resampled by the device kernel instead of on the host (a 10 s
    44.1 kHz file: 0.1 ms instead of 21 ms).  With librosa installed the host path of ``load`` is used unchanged, so that
    decoding equals the reference's.  Returns a float32 tensor [N] on ``device``."""
    import torch
    try:
        import librosa  # noqa: F401
        return torch.from_numpy(load(path, sr)[0]).to(device)
    except ImportError:
        pass
    x, rate = read_native(path)
    mono = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    y = torch.from_numpy(np.ascontiguousarray(mono, dtype=np.float32)).to(device)
    return y if sr is None or int(rate) == int(sr) else resample_on_device(y, rate, sr)
