"""MPEG-1 Audio Layer III decoder, written from the standard's decoding process (ISO/IEC 11172-3, 2.4.3.4): the decode half
of ``librosa.load(path)`` for the reference's MP3 inputs (reference call sites: openvoice/api.py:123,144;
BASELINE.json configs[0] names resources/example_reference.mp3).  Pure numpy, host side only -- file decoding is not on the
hot path (it happens once per utterance, before the waveform reaches the GPU).

    pcm, rate = decode(open(path, "rb").read())        # float32 [channels, samples], the file's own sampling rate

Scope: Layer III of MPEG-1 (32 / 44.1 / 48 kHz), MPEG-2 (16 / 22.05 / 24 kHz: the LSF extension of ISO/IEC 13818-3 -- what
24 kHz text-to-speech services return) and MPEG-2.5 (8 / 11.025 / 12 kHz); mono / stereo / joint stereo with MS and
intensity coding, long / short / mixed blocks, bit reservoir, CRC-protected frames (the CRC is skipped, not checked), ID3v2
tags, the Xing / Info header frame with LAME's gapless fields (encoder delay and padding are trimmed the way FFmpeg trims
them).  NOT built: mixed blocks at 8 kHz (``Mp3Error`` names them) -- the one place where the standard's two rules
disagree (long bands 0..5 are 72 lines there, the long-windowed part of a mixed block 36) and FFmpeg, the decoder of the
reference's chain, has no defined behaviour either (it asks for a sample file).

Tables: the standard's Huffman code tables (Annex B, Table B.7), synthesis window (Table B.3) and scalefactor-band
partitions (Table B.8) are normative data that no formula produces; ``mp3_tables.npz`` holds them, read out of the image's
bundled Chromium / FFmpeg build and validated structurally by tools/extract_mp3_tables.py.  Everything else -- IMDCT and
window shapes, alias-reduction butterflies, the synthesis matrix, requantisation -- is computed here from its definition.

Pinned: tests/test_mp3_cpu.py compares the output with Chromium's (FFmpeg's) decode of the same files
(oracle/make_mp3_golden.py -> tests/golden/mp3_*.npz)."""
import os

import numpy as np


class Mp3Error(ValueError):
    pass


_T = None                       # tables, loaded on first use
BITRATES = (0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)          # kbit/s, MPEG-1 Layer III
BITRATES_LSF = (0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160)           # MPEG-2 / 2.5 Layer III
RATES = (44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000)                  # table row = 3 * family + index
FAMILY = {3: 0, 2: 1, 0: 2}        # header version bits -> MPEG-1, MPEG-2 (LSF), MPEG-2.5 (LSF at a quarter of the rates)
# MPEG-2 LSF scale factors (ISO/IEC 13818-3, 2.4.3.2): bands per partition, by scalefac_compress range and block kind
# (long, short, mixed); partition i is coded with slen[i] bits per factor
NR_OF_SFB = (((6, 5, 5, 5), (9, 9, 9, 9), (6, 9, 9, 9)),
             ((6, 5, 7, 3), (9, 9, 12, 6), (6, 9, 12, 6)),
             ((11, 10, 0, 0), (18, 18, 0, 0), (15, 18, 0, 0)),
             # the right channel of an intensity-stereo frame (its factors are intensity positions above the bound):
             ((7, 7, 7, 0), (12, 12, 12, 0), (6, 15, 12, 0)),
             ((6, 6, 6, 3), (12, 9, 9, 6), (6, 12, 9, 6)),
             ((8, 8, 5, 0), (15, 12, 9, 0), (6, 18, 9, 0)))
LINBITS = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13)
TABLE_OF = (0, 1, 2, 3, 0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 0, 15) + (16,) * 8 + (24,) * 8   # table_select -> code table
SLEN = ((0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4), (0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3))
PRETAB = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0)
ALIAS_C = (-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037)


def _tables():
    """Everything derived once: Huffman look-up lists (peek max-length bits -> length, x, y), the per-rate line -> band
    maps, the IMDCT matrices with their windows folded in, the synthesis matrix and window."""
    global _T
    if _T is not None:
        return _T
    raw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mp3_tables.npz"))
    t = {}
    huff = {}
    for tid in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 24):
        lens, codes = raw[f"huff{tid}_len"], raw[f"huff{tid}_code"]
        maxlen = int(lens.max())
        lut = [0] * (1 << maxlen)                      # entry: len << 8 | x << 4 | y
        for x in range(lens.shape[0]):
            for y in range(lens.shape[1]):
                n, c = int(lens[x, y]), int(codes[x, y])
                lo = c << (maxlen - n)
                lut[lo:lo + (1 << (maxlen - n))] = [n << 8 | x << 4 | y] * (1 << (maxlen - n))
        huff[tid] = (maxlen, lut)
    t["huff"] = huff
    quad = []
    for q in range(2):
        lens, codes = raw["count1_len"][q], raw["count1_code"][q]
        maxlen = int(lens.max())
        lut = [0] * (1 << maxlen)
        for v in range(16):
            n, c = int(lens[v]), int(codes[v])
            lo = c << (maxlen - n)
            lut[lo:lo + (1 << (maxlen - n))] = [n << 8 | v] * (1 << (maxlen - n))
        quad.append((maxlen, lut))
    t["quad"] = quad
    # scalefactor bands (Table B.8): boundaries in lines (long) / lines of one window (short)
    t["long_idx"] = [np.concatenate([[0], np.cumsum(w)]).astype(int) for w in raw["sfb_long_width"]]
    t["short_idx"] = [np.concatenate([[0], np.cumsum(w)]).astype(int) for w in raw["sfb_short_width"]]
    t["long_of_line"], t["short_of_line"], t["short_win_of_line"], t["reorder"], t["reorder_mixed"] = [], [], [], [], []
    for r in range(9):
        li, si = t["long_idx"][r], t["short_idx"][r]
        t["long_of_line"].append(np.repeat(np.arange(22), np.diff(li)))
        # short blocks as transmitted: band by band, inside a band window 0's lines, then window 1's, window 2's
        band = np.concatenate([np.full(3 * (si[b + 1] - si[b]), b) for b in range(13)])
        win = np.concatenate([np.repeat(np.arange(3), si[b + 1] - si[b]) for b in range(13)])
        t["short_of_line"].append(band)
        t["short_win_of_line"].append(win)
        # destination of transmitted line n for the IMDCT: 3 * (frequency line inside the window) + window
        dst = np.concatenate([(3 * (si[b] + np.arange(si[b + 1] - si[b]))[None, :] + np.arange(3)[:, None]).reshape(-1)
                              for b in range(13)])
        t["reorder"].append(dst)
        mixed = np.arange(576)
        mixed[36:] = dst[36:]                          # the first two subbands (36 lines = short bands 0..2) stay long
        # (8 kHz: short bands 0..2 are 72 lines wide; mixed blocks there are not decoded, see decode())
        t["reorder_mixed"].append(mixed)
    t["pow43"] = np.arange(8207, dtype=np.float64) ** (4.0 / 3.0)
    c = np.array(ALIAS_C)
    t["cs"], t["ca"] = 1.0 / np.sqrt(1.0 + c * c), c / np.sqrt(1.0 + c * c)
    # IMDCT, n = 36: x[i] = sum_k X[k] cos(pi / 72 (2 i + 1 + 18) (2 k + 1)), times the block-type window
    i, k = np.arange(36)[:, None], np.arange(18)[None, :]
    c36 = np.cos(np.pi / 72.0 * (2 * i + 1 + 18) * (2 * k + 1))
    ii = np.arange(36)
    w0 = np.sin(np.pi / 36.0 * (ii + 0.5))
    w1 = np.where(ii < 18, w0, np.where(ii < 24, 1.0, np.where(ii < 30, np.sin(np.pi / 12.0 * (ii - 18 + 0.5)), 0.0)))
    w3 = np.where(ii < 6, 0.0, np.where(ii < 12, np.sin(np.pi / 12.0 * (ii - 6 + 0.5)), np.where(ii < 18, 1.0, w0)))
    t["imdct36"] = {0: (c36 * w0[:, None]).T, 1: (c36 * w1[:, None]).T, 3: (c36 * w3[:, None]).T}      # [18][36]
    i, k = np.arange(12)[:, None], np.arange(6)[None, :]
    c12 = np.cos(np.pi / 24.0 * (2 * i + 1 + 6) * (2 * k + 1)) * np.sin(np.pi / 12.0 * (np.arange(12) + 0.5))[:, None]
    # three windows of a short block laid into the 36-sample frame at 6 + 6 w: lines 3 k + w of a subband -> 36 samples
    short = np.zeros((18, 36))
    for w in range(3):
        for kk in range(6):
            short[3 * kk + w, 6 + 6 * w:18 + 6 * w] = c12[:, kk]
    t["imdct_short"] = short                                                                           # [18][36]
    # polyphase synthesis: V[i] = sum_k cos((16 + i)(2 k + 1) pi / 64) S[k]; window D[i] = C[i] * 32
    i, k = np.arange(64)[:, None], np.arange(32)[None, :]
    t["synth"] = np.cos((16 + i) * (2 * k + 1) * np.pi / 64.0).T                                       # [32][64]
    half = raw["window_half"].astype(np.float64) / 65536.0
    d = np.zeros(512)
    d[:257] = half                                   # (the table already alternates in sign from one 64-tap block to the next)
    for n in range(1, 256):                          # second half: D[512 - n] = -D[n], except D[512 - 64 k] = D[64 k]
        d[512 - n] = -half[n] if n % 64 else half[n]
    t["window"] = d
    _T = t
    return t


_CORE = None                                            # ctypes handle of libov_mp3.so; False = not built


def _bit(buf, pos):
    """Bit ``pos`` of ``buf``; zero beyond its end (a corrupt big_values / linbits field runs past the granule's data:
    the C core reads zeros there too, so both forms decode damaged input identically instead of raising IndexError)."""
    b = pos >> 3
    return (buf[b] >> (7 - (pos & 7))) & 1 if b < len(buf) else 0


MP3_CORE_VERSION = 1          # ovmp3_version() of csrc/mp3_core.c this module was written against


def _core():
    """The C Huffman core (csrc/mp3_core.c -> libov_mp3.so) when it is built, else None (the Python form below is used).
    ``OPENVOICE_AMD_MP3_CORE=0`` forces the Python form (tests compare the two)."""
    global _CORE
    if os.environ.get("OPENVOICE_AMD_MP3_CORE", "1") == "0":
        return None
    if _CORE is None:
        import ctypes
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libov_mp3.so")
        try:
            lib = ctypes.CDLL(path)
            lib.ovmp3_version.restype, lib.ovmp3_version.argtypes = ctypes.c_int, []
            if lib.ovmp3_version() != MP3_CORE_VERSION:      # a stale libov_mp3.so: the Python form is the safe choice
                raise OSError(f"libov_mp3.so is version {lib.ovmp3_version()}, expected {MP3_CORE_VERSION}")
            lib.ovmp3_huffman.restype = ctypes.c_int64
            lib.ovmp3_huffman.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p]
            _CORE = lib
        except (OSError, AttributeError):
            _CORE = False
    return _CORE or None


def _huffman_core(lib, br, g, end, bounds, t):
    """``_huffman`` through the C core: the same look-up tables as uint32 arrays, the same bit positions."""
    import ctypes
    c = t.get("core")
    if c is None:
        c = t["core"] = dict(
            lut={tid: np.asarray(lut, dtype=np.uint32) for tid, (_, lut) in t["huff"].items()},
            quad=[np.asarray(lut, dtype=np.uint32) for _, lut in t["quad"]])
    ptrs = (ctypes.c_void_p * 3)()
    ml, lb = np.zeros(3, dtype=np.int32), np.zeros(3, dtype=np.int32)
    for region in range(3):
        sel = g["table_select"][region]
        tid = TABLE_OF[sel]
        if tid:
            ptrs[region] = c["lut"][tid].ctypes.data
            ml[region], lb[region] = t["huff"][tid][0], LINBITS[sel]
    q = g["count1table_select"]
    out = np.empty(578, dtype=np.int32)
    b = np.asarray(bounds, dtype=np.int32)
    lib.ovmp3_huffman(br.buf, len(br.buf), br.pos, end, b.ctypes.data, ctypes.addressof(ptrs), ml.ctypes.data, lb.ctypes.data,
                      c["quad"][q].ctypes.data, t["quad"][q][0], out.ctypes.data)
    br.pos = end
    return out[:576]


class _Bits:
    """MSB-first bit reader over a bytes-like object."""
    __slots__ = ("buf", "pos", "n")

    def __init__(self, buf, pos=0):
        self.buf, self.pos, self.n = buf, pos, len(buf) * 8

    def get(self, n):
        if n == 0:
            return 0
        p = self.pos
        self.pos = p + n
        b = p >> 3
        chunk = int.from_bytes(self.buf[b:b + 5], "big") << (8 * (5 - len(self.buf[b:b + 5])))
        return (chunk >> (40 - (p & 7) - n)) & ((1 << n) - 1)

    def peek(self, n):
        p = self.pos
        b = p >> 3
        piece = self.buf[b:b + 5]
        chunk = int.from_bytes(piece, "big") << (8 * (5 - len(piece)))
        return (chunk >> (40 - (p & 7) - n)) & ((1 << n) - 1)


def _skip_id3v2(data):
    pos = 0
    while data[pos:pos + 3] == b"ID3" and len(data) >= pos + 10:
        size = (data[pos + 6] & 0x7F) << 21 | (data[pos + 7] & 0x7F) << 14 | (data[pos + 8] & 0x7F) << 7 | (data[pos + 9] & 0x7F)
        pos += 10 + size + (10 if data[pos + 5] & 0x10 else 0)
    return pos


def _header(data, pos):
    """Parsed frame header at ``pos`` or None."""
    if pos + 4 > len(data):
        return None
    h = int.from_bytes(data[pos:pos + 4], "big")
    if (h >> 21) != 0x7FF:
        return None
    version, layer, prot = (h >> 19) & 3, (h >> 17) & 3, (h >> 16) & 1
    bri, sri, pad = (h >> 12) & 15, (h >> 10) & 3, (h >> 9) & 1
    if version == 1 or layer == 0 or bri == 15 or sri == 3:
        return None
    fam = FAMILY[version]
    return dict(version=version, layer=layer, crc=prot == 0, bitrate_index=bri, rate_index=sri, padding=pad,
                mode=(h >> 6) & 3, mode_ext=(h >> 4) & 3, lsf=fam > 0, row=3 * fam + sri)


def _frame_length(hd):
    """Bytes of the frame: 1152 samples per frame for MPEG-1, 576 (one granule) for the LSF families."""
    if hd["lsf"]:
        return 72000 * BITRATES_LSF[hd["bitrate_index"]] // RATES[hd["row"]] + hd["padding"]
    return 144000 * BITRATES[hd["bitrate_index"]] // RATES[hd["row"]] + hd["padding"]


def _side_len(hd):
    mono = hd["mode"] == 3
    return (9 if mono else 17) if hd["lsf"] else (17 if mono else 32)


def _same_stream(a, b):
    """Two headers of one stream agree in version, layer and sampling rate (bit rate, padding, mode may change)."""
    return a["version"] == b["version"] and a["layer"] == b["layer"] and a["rate_index"] == b["rate_index"]


def _confirmed(data, pos, hd):
    """A byte pattern that parses as a header is a FRAME only when the stream continues consistently behind it: another
    header of the same version / layer / rate at ``pos + frame length``, an ID3v1 ``TAG`` there, or the end of the data
    (the last frame).  Anything else -- cover art, an APEv2 tag, a trailer, a damaged header -- is skipped byte by byte,
    as FFmpeg's resynchronisation does, instead of aborting the file."""
    if hd is None or hd["bitrate_index"] == 0 or hd["layer"] != 1:
        # (free-format streams and Layers I / II are not frames to this decoder; their length formulas differ, so they
        # are confirmed -- for the error message -- by _first_frame's own rule below)
        return False
    end = pos + _frame_length(hd)
    if end >= len(data) - 4:
        return end <= len(data)
    if data[end:end + 3] == b"TAG":
        return True
    nxt = _header(data, end)
    return nxt is not None and _same_stream(hd, nxt)


def _layer12_frame_length(hd):
    """Frame bytes of a Layer I / II header (only to CONFIRM such a stream before naming it in the error)."""
    v1 = hd["version"] == 3
    rate = RATES[hd["row"]]
    if hd["layer"] == 3:                                   # Layer I: 32-bit slots
        kbps = ((0, 32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448) if v1 else
                (0, 32, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 256))[hd["bitrate_index"]]
        return (12000 * kbps // rate + hd["padding"]) * 4
    kbps = ((0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384) if v1 else
            (0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160))[hd["bitrate_index"]]
    return 144000 * kbps // rate + hd["padding"]


def _first_frame(data):
    pos = _skip_id3v2(data)
    while pos + 4 <= len(data):
        hd = _header(data, pos)
        if hd is not None and hd["bitrate_index"] != 0:
            if hd["layer"] == 1:
                if _confirmed(data, pos, hd):
                    return pos, hd
            else:
                # Layer I / II: named only when a second consistent header confirms it is a stream, not noise
                nxt = _header(data, pos + _layer12_frame_length(hd))
                if nxt is not None and _same_stream(hd, nxt):
                    raise Mp3Error("only Layer III is built (this stream is MPEG audio Layer I or II)")
        pos += 1
    raise Mp3Error("unsupported format: no MPEG audio Layer III frame found (RIFF/WAVE and MP3 are read natively)")


def probe(data):
    """Sampling rate, channel count and the gapless fields of the Xing / Info + LAME header (if any)."""
    pos, hd = _first_frame(data)
    info = dict(sample_rate=RATES[hd["row"]], channels=1 if hd["mode"] == 3 else 2, first_frame=pos, xing=False,
                frames=None, start_pad=0, end_pad=0, lsf=hd["lsf"])
    side = _side_len(hd)
    tag = pos + 4 + side                              # (the header frame carries no CRC in practice; FFmpeg assumes so too)
    if data[tag:tag + 4] in (b"Xing", b"Info"):
        info["xing"] = True
        flags = int.from_bytes(data[tag + 4:tag + 8], "big")
        p = tag + 8
        if flags & 1:
            info["frames"] = int.from_bytes(data[p:p + 4], "big")
            p += 4
        if flags & 2:
            p += 4
        if flags & 4:
            p += 100
        if flags & 8:
            p += 4
        # LAME extension: 9-byte encoder string, 12 bytes of settings, then delay (12 bits) | padding (12 bits)
        if p + 24 <= len(data) and data[p:p + 4] in (b"LAME", b"Lavc", b"Lavf", b"L3.9"):
            v = int.from_bytes(data[p + 21:p + 24], "big")
            info["start_pad"], info["end_pad"] = v >> 12, v & 0xFFF
    return info


def _side_info(br, nch):
    si = dict(main_data_begin=br.get(9))
    br.get(5 if nch == 1 else 3)
    si["scfsi"] = [[br.get(1) for _ in range(4)] for _ in range(nch)]
    si["gr"] = []
    for _ in range(2):
        chans = []
        for _ in range(nch):
            g = dict(part2_3_length=br.get(12), big_values=br.get(9), global_gain=br.get(8), scalefac_compress=br.get(4),
                     window_switching=br.get(1))
            if g["window_switching"]:
                g["block_type"], g["mixed"] = br.get(2), br.get(1)
                g["table_select"] = [br.get(5), br.get(5), 0]
                g["subblock_gain"] = [br.get(3), br.get(3), br.get(3)]
                if g["block_type"] == 0:
                    raise Mp3Error("invalid stream: window switching with block type 0")
                g["region0_count"], g["region1_count"] = (8 if g["block_type"] == 2 and not g["mixed"] else 7), 36
            else:
                g["block_type"], g["mixed"] = 0, 0
                g["table_select"] = [br.get(5), br.get(5), br.get(5)]
                g["subblock_gain"] = [0, 0, 0]
                g["region0_count"], g["region1_count"] = br.get(4), br.get(3)
            g["preflag"], g["scalefac_scale"], g["count1table_select"] = br.get(1), br.get(1), br.get(1)
            chans.append(g)
        si["gr"].append(chans)
    return si


def _side_info_lsf(br, nch):
    """MPEG-2 / 2.5 side information (ISO/IEC 13818-3, 2.4.1.7): one granule per frame, no scfsi, 9-bit
    scalefac_compress, no preflag bit (it follows from scalefac_compress)."""
    si = dict(main_data_begin=br.get(8))
    br.get(1 if nch == 1 else 2)
    chans = []
    for _ in range(nch):
        g = dict(part2_3_length=br.get(12), big_values=br.get(9), global_gain=br.get(8), scalefac_compress=br.get(9),
                 window_switching=br.get(1))
        if g["window_switching"]:
            g["block_type"], g["mixed"] = br.get(2), br.get(1)
            g["table_select"] = [br.get(5), br.get(5), 0]
            g["subblock_gain"] = [br.get(3), br.get(3), br.get(3)]
            if g["block_type"] == 0:
                raise Mp3Error("invalid stream: window switching with block type 0")
            g["region0_count"], g["region1_count"] = (8 if g["block_type"] == 2 and not g["mixed"] else 7), 36
        else:
            g["block_type"], g["mixed"] = 0, 0
            g["table_select"] = [br.get(5), br.get(5), br.get(5)]
            g["subblock_gain"] = [0, 0, 0]
            g["region0_count"], g["region1_count"] = br.get(4), br.get(3)
        g["scalefac_scale"], g["count1table_select"] = br.get(1), br.get(1)
        g["preflag"] = 1 if g["scalefac_compress"] >= 500 else 0
        chans.append(g)
    si["scfsi"] = [[0, 0, 0, 0] for _ in range(nch)]
    si["gr"] = [chans]
    return si


def _scalefactors_lsf(br, g, intensity_right=False):
    """LSF scale factors: scalefac_compress selects four bit widths and, with the block kind, how many bands each
    partition holds (NR_OF_SFB); the factors follow band by band (short blocks: the three windows of a band in turn).
    ``intensity_right``: the right channel of an intensity-stereo frame -- bit 0 of scalefac_compress is the intensity
    scale, the rest selects among three other partitions (ISO/IEC 13818-3, 2.4.3.2), and there is no preflag."""
    sfc = g["scalefac_compress"]
    if intensity_right:
        sfc >>= 1
        g["preflag"] = 0
        if sfc < 180:
            slen, row = (sfc // 36, (sfc % 36) // 6, (sfc % 36) % 6, 0), 3
        elif sfc < 244:
            sfc -= 180
            slen, row = ((sfc & 63) >> 4, (sfc & 15) >> 2, sfc & 3, 0), 4
        else:
            sfc -= 244
            slen, row = (sfc // 3, sfc % 3, 0, 0), 5
    elif sfc < 400:
        slen, row = ((sfc >> 4) // 5, (sfc >> 4) % 5, (sfc % 16) >> 2, sfc % 4), 0
    elif sfc < 500:
        sfc -= 400
        slen, row = ((sfc >> 2) // 5, (sfc >> 2) % 5, sfc % 4, 0), 1
    else:
        sfc -= 500
        slen, row = (sfc // 3, sfc % 3, 0, 0), 2
    kind = 0 if g["block_type"] != 2 else (2 if g["mixed"] else 1)
    vals = []
    for n, bits in zip(NR_OF_SFB[row][kind], slen):
        vals += [br.get(bits) for _ in range(n)]
    long_sf, short_sf = [0] * 22, [[0, 0, 0] for _ in range(13)]
    if kind == 0:
        long_sf[:len(vals)] = vals
    else:
        first = 0
        if kind == 2:                                   # mixed: 6 long bands, then the short bands from band 3 on
            long_sf[:6] = vals[:6]
            vals, first = vals[6:], 3
        for i, v in enumerate(vals):
            short_sf[first + i // 3][i % 3] = v
    return long_sf, short_sf


def _scalefactors(br, g, scfsi, prev_long):
    """Scale factors of one (granule, channel): (long [22], short [13][3]); ``prev_long`` = granule 0's long factors of
    this channel when scfsi may reuse them (granule 1), else None.  ISO 11172-3 2.4.2.7 / 2.4.3.4.5."""
    slen1, slen2 = SLEN[0][g["scalefac_compress"]], SLEN[1][g["scalefac_compress"]]
    long_sf, short_sf = [0] * 22, [[0, 0, 0] for _ in range(13)]
    if g["block_type"] == 2:
        if g["mixed"]:
            for sfb in range(8):
                long_sf[sfb] = br.get(slen1)
            for sfb in range(3, 6):
                for w in range(3):
                    short_sf[sfb][w] = br.get(slen1)
        else:
            for sfb in range(6):
                for w in range(3):
                    short_sf[sfb][w] = br.get(slen1)
        for sfb in range(6, 12):
            for w in range(3):
                short_sf[sfb][w] = br.get(slen2)
    else:
        for grp, (lo, hi) in enumerate(((0, 6), (6, 11), (11, 16), (16, 21))):
            n = slen1 if grp < 2 else slen2
            if prev_long is not None and scfsi[grp]:
                long_sf[lo:hi] = prev_long[lo:hi]
            else:
                for sfb in range(lo, hi):
                    long_sf[sfb] = br.get(n)
    return long_sf, short_sf


def _huffman(br, g, end, rate_index, t):
    """576 quantised lines of one (granule, channel); the reader ends at bit ``end`` (= part 2 start + part2_3_length)."""
    out = [0] * 578
    big = min(g["big_values"] * 2, 576)
    li = t["long_idx"][rate_index]
    if g["window_switching"]:
        # region 0 ends after short band 2 (3 x 12 lines) / long band 7 -- both 36 lines at the MPEG-1 rates, 54 / 36 at the
        # LSF rates' band tables (and 108 / 72 at 8 kHz); region 1 takes the rest
        r1, r2 = (3 * int(t["short_idx"][rate_index][3]) if g["block_type"] == 2 else int(li[8])), 576
    else:
        r1 = int(li[min(g["region0_count"] + 1, 22)])
        r2 = int(li[min(g["region0_count"] + g["region1_count"] + 2, 22)])
    bounds = (min(r1, big), min(r2, big), big)
    lib = _core()
    if lib is not None:
        return _huffman_core(lib, br, g, end, bounds, t)
    buf, pos = br.buf, br.pos
    i = 0
    for region in range(3):
        sel = g["table_select"][region]
        tid, linbits, stop = TABLE_OF[sel], LINBITS[sel], bounds[region]
        if tid == 0:
            i = max(i, stop)                           # table 0: all zeros, no bits
            continue
        maxlen, lut = t["huff"][tid]
        shift, mask = 40 - maxlen, (1 << maxlen) - 1
        while i < stop:
            b = pos >> 3
            piece = buf[b:b + 5]
            word = int.from_bytes(piece, "big") << (8 * (5 - len(piece)))
            e = lut[(word >> (shift - (pos & 7))) & mask]
            pos += e >> 8
            x, y = (e >> 4) & 15, e & 15
            if linbits and x == 15:
                br.pos = pos
                x += br.get(linbits)
                pos = br.pos
            if x:
                if _bit(buf, pos):
                    x = -x
                pos += 1
            if linbits and y == 15:
                br.pos = pos
                y += br.get(linbits)
                pos = br.pos
            if y:
                if _bit(buf, pos):
                    y = -y
                pos += 1
            out[i], out[i + 1] = x, y
            i += 2
    # count1 region: quadruples of magnitude <= 1 until the granule's bits run out
    maxlen, lut = t["quad"][g["count1table_select"]]
    while pos < end and i <= 572:
        b = pos >> 3
        piece = buf[b:b + 5]
        word = int.from_bytes(piece, "big") << (8 * (5 - len(piece)))
        e = lut[(word >> (40 - maxlen - (pos & 7))) & ((1 << maxlen) - 1)]
        pos += e >> 8
        v = e & 15
        vals = [0, 0, 0, 0]
        for n, bit in enumerate((8, 4, 2, 1)):
            if v & bit:
                vals[n] = -1 if _bit(buf, pos) else 1
                pos += 1
        if pos > end:                                  # ran past the granule: the last quadruple is stuffing, not data
            break
        out[i:i + 4] = vals
        i += 4
    br.pos = end
    return out[:576]


def _requantise(isamp, g, long_sf, short_sf, rate_index, t):
    """xr = sign(is) |is|^(4/3) 2^((global_gain - 210) / 4) x the scale-factor term (2.4.3.4.7.1), as transmitted
    (short blocks not yet reordered)."""
    isamp = np.asarray(isamp, dtype=np.int64)
    mag = t["pow43"][np.minimum(np.abs(isamp), 8206)] * np.sign(isamp)
    mult = 1.0 if g["scalefac_scale"] else 0.5
    gain = (g["global_gain"] - 210) / 4.0
    lsf = np.asarray(long_sf, dtype=np.float64)
    exp_long = gain - mult * (lsf + (np.asarray(PRETAB) if g["preflag"] else 0))[t["long_of_line"][rate_index]]
    if g["block_type"] != 2:
        return mag * np.exp2(exp_long)
    ssf = np.asarray(short_sf, dtype=np.float64)
    band, win = t["short_of_line"][rate_index], t["short_win_of_line"][rate_index]
    exp_short = gain - 2.0 * np.asarray(g["subblock_gain"], dtype=np.float64)[win] - mult * ssf[band, win]
    if g["mixed"]:
        exp_short[:36] = exp_long[:36]
    return mag * np.exp2(exp_short)


def _joint_stereo(xr, nonzero_right, g1, long_sf1, short_sf1, row, t, ms, lsf):
    """Intensity stereo (ISO/IEC 11172-3, 2.4.3.4.9.3; LSF: 13818-3, 2.4.3.2) with MS stereo below the bound, on the two
    channels' lines as transmitted.  The right channel's block type ``g1`` gives the band partition; in every window the
    bands ABOVE the last band in which the right channel transmitted a non-zero line carry, in the left channel, the sum
    signal, and in the right channel's scale factors the intensity position.  A band with an illegal position (>= 7;
    LSF: >= 16, FFmpeg's reading -- the decoder of the reference's chain) is not intensity coded: MS if the frame says
    MS, else as transmitted.  The last band (long 21, short 12) has no factor of its own and takes the one below."""
    nz = np.asarray(nonzero_right, dtype=bool)
    region = np.zeros(576, dtype=bool)                  # lines coded in intensity stereo
    pos = np.zeros(576, dtype=np.int64)
    lband = t["long_of_line"][row]
    lsf1 = np.asarray(long_sf1, dtype=np.int64)
    if g1["block_type"] == 2:
        sband, swin = t["short_of_line"][row], t["short_win_of_line"][row]
        first = 36 if g1["mixed"] else 0                # mixed: the lines below are long bands
        ssf = np.asarray(short_sf1, dtype=np.int64)
        found_any = False
        for w in range(3):
            sel = (swin == w) & (np.arange(576) >= first)
            hit = sband[sel & nz]
            top = int(hit.max()) if hit.size else -1
            found_any |= hit.size > 0
            region |= sel & (sband > top)
        short_lines = np.arange(576) >= first
        pos[short_lines] = ssf[np.minimum(sband, 11), swin][short_lines]
        if g1["mixed"]:
            low = np.arange(576) < 36
            if not found_any:
                hit = lband[low & nz]
                top = int(hit.max()) if hit.size else -1
                region |= low & (lband > top)
            pos[low] = lsf1[lband][low]
    else:
        hit = lband[nz]
        top = int(hit.max()) if hit.size else -1
        region = lband > top
        pos = lsf1[np.minimum(lband, 20)]
    if lsf:
        # position p: the channel (left for odd p, right for even p) is scaled by 2^(-(scale + 1) ((p + 1) >> 1) / 4)
        scale = g1["scalefac_compress"] & 1
        legal = pos < 16
        p = np.minimum(pos, 15)
        f = np.exp2(-(scale + 1) * ((p + 1) >> 1) / 4.0)
        left, right = np.where(p & 1, f, 1.0), np.where(p & 1, 1.0, f)
    else:
        legal = pos < 7
        ratio = np.tan(np.minimum(pos, 5) * np.pi / 12.0)
        left = np.where(pos == 6, 1.0, ratio / (1.0 + ratio))
        right = np.where(pos == 6, 0.0, 1.0 / (1.0 + ratio))
    intensity = region & legal
    x0, x1 = xr[0].copy(), xr[1].copy()
    if ms:
        xr[0], xr[1] = (x0 + x1) / np.sqrt(2.0), (x0 - x1) / np.sqrt(2.0)
    xr[0][intensity] = (x0 * left)[intensity]
    xr[1][intensity] = (x0 * right)[intensity]


HYBRID_BLOCK = 64


def _hybrid(x, kind, tail, row, t):
    """Hybrid synthesis of a block of granules of one channel: ``x`` [n][576] lines as transmitted, ``kind`` [n] (0 / 1 / 3
    long with the normal / start / stop window, 2 short, 4 mixed, -1 undecodable = silence), ``tail`` [32][18] the overlap
    left by the granule before.  Returns ([18 n slots][32 subbands], new tail)."""
    n = x.shape[0]
    for code, key in ((2, "reorder"), (4, "reorder_mixed")):                # short blocks: band-by-band order -> 3 k + window
        sel = kind == code                                                  # inside a subband
        if sel.any():
            y = np.zeros((int(sel.sum()), 576))
            y[:, t[key][row]] = x[sel]
            x[sel] = y
    x = x.reshape(n, 32, 18)
    for sel, nlong in (((kind == 0) | (kind == 1) | (kind == 3), 32), (kind == 4, 2)):
        if sel.any():                                  # alias reduction across the boundaries between LONG subbands: the
            # butterflies of different boundaries touch disjoint lines (top 8 of sb - 1, bottom 8 of sb): all at once
            xs = x[sel]
            a, b = xs[:, :nlong - 1, 17:9:-1].copy(), xs[:, 1:nlong, :8].copy()
            xs[:, :nlong - 1, 17:9:-1] = a * t["cs"] - b * t["ca"]
            xs[:, 1:nlong, :8] = b * t["cs"] + a * t["ca"]
            x[sel] = xs
    out = np.zeros((n, 32, 36))
    mm = lambda a, m: (np.ascontiguousarray(a).reshape(-1, 18) @ m).reshape(a.shape[0], -1, 36)   # one GEMM per window kind
    for bt in (0, 1, 3):
        sel = kind == bt
        if sel.any():
            out[sel] = mm(x[sel], t["imdct36"][bt])
    sel = kind == 2
    if sel.any():
        out[sel] = mm(x[sel], t["imdct_short"])
    sel = kind == 4
    if sel.any():
        xs = x[sel]
        out[sel] = np.concatenate([mm(xs[:, :2], t["imdct36"][0]), mm(xs[:, 2:], t["imdct_short"])], axis=1)
    tails = np.concatenate([tail[None], out[:, :, 18:]], axis=0)            # tails[i] = what granule i overlaps with
    silent = np.flatnonzero(kind == -1)
    for i in silent:                                   # an undecodable granule passes the overlap before it on
        tails[i + 1] = tails[i]
    hyb = out[:, :, :18] + tails[:-1]
    hyb[silent] = 0.0
    hyb[:, 1::2, 1::2] *= -1.0                         # frequency inversion: odd time samples of odd subbands
    return hyb.transpose(0, 2, 1).reshape(-1, 32), tails[-1]


def decode(data, trim_gapless=True, clip=True):
    """``(pcm float32 [channels, samples], sampling rate)`` of an MPEG-1 Layer III stream held in ``data`` (bytes).
    ``trim_gapless``: drop the encoder delay / padding announced in a LAME header the way FFmpeg does; ``clip``: saturate
    at +-1 like the 16-bit PCM the reference's decode chain goes through (librosa -> audioread -> FFmpeg; Chromium's
    FFmpeg, the pin of this decoder, saturates the same way -- demo_speaker1.mp3 peaks at 1.22 unclipped)."""
    t = _tables()
    data = bytes(data)
    info = probe(data)
    pos, nch, rate = info["first_frame"], info["channels"], info["sample_rate"]
    reservoir = b""
    # per channel, one entry per granule: the 576 requantised (and stereo-processed) lines as transmitted, and what the
    # hybrid filter bank has to do with them -- 0 / 1 / 3 long blocks (normal / start / stop window), 2 short, 4 mixed,
    # -1 a granule that could not be decoded (silence).  The filter bank itself runs once, over all granules (below).
    lines = [[] for _ in range(nch)]
    kinds = [[] for _ in range(nch)]
    first = True
    frames = 0
    stream_hd = _header(data, pos)                        # the confirmed first frame: version / layer / rate of the stream
    in_sync = -1                                          # where the frame after the last decoded one must start
    while True:
        hd = _header(data, pos)
        # in sync (the header sits exactly where the previous frame ended) a frame of this stream is taken as is -- so the
        # last frame before a trailer is not lost; out of sync it must be confirmed by what follows it
        ok = hd is not None and hd["bitrate_index"] != 0 and _same_stream(hd, stream_hd) and (
            pos == in_sync or _confirmed(data, pos, hd))
        if not ok:
            # resynchronise: junk between frames (tags, cover art, a damaged header) is skipped to the next CONFIRMED
            # frame of this stream; stop when nothing follows
            nxt = data.find(b"\xff", pos + 1)
            while nxt >= 0:
                cand = _header(data, nxt)
                if cand is not None and _same_stream(cand, stream_hd) and _confirmed(data, nxt, cand):
                    break
                nxt = data.find(b"\xff", nxt + 1)
            if nxt < 0:
                break
            pos = nxt
            continue
        flen = _frame_length(hd)
        if pos + flen > len(data):
            break
        frame_nch = 1 if hd["mode"] == 3 else 2
        ngr, row = (1 if hd["lsf"] else 2), hd["row"]
        if frame_nch != nch and not (first and info["xing"]):
            # a frame of this stream whose channel count differs from the first frame's -- in practice a damaged mode
            # field: its granules are silence and the stream goes on (nothing inside the stream aborts the file; only
            # the confirmed FIRST frame decides what the stream is)
            for ch in range(nch):
                for _ in range(ngr):
                    lines[ch].append(np.zeros(576))
                    kinds[ch].append(-1)
            first = False
            pos += flen
            in_sync = pos
            frames += 1
            continue
        side_len = _side_len(hd)
        body = pos + 4 + (2 if hd["crc"] else 0)
        if first and info["xing"]:
            first = False                              # the Xing / Info frame carries the header, not audio
            pos += flen
            in_sync = pos
            continue
        first = False
        in_sync = pos + flen
        si = (_side_info_lsf if hd["lsf"] else _side_info)(_Bits(data, body * 8), nch)
        main = data[body + side_len:pos + flen]
        if si["main_data_begin"] > len(reservoir):
            # the stream was cut before this frame's reservoir: its granules cannot be decoded -> silence (as FFmpeg)
            reservoir = (reservoir + main)[-511:]
            for ch in range(nch):
                for _ in range(ngr):
                    lines[ch].append(np.zeros(576))
                    kinds[ch].append(-1)
            pos += flen
            frames += 1
            continue
        buf = reservoir[len(reservoir) - si["main_data_begin"]:] + main
        reservoir = (reservoir + main)[-511:]
        br = _Bits(buf + bytes(8))                     # (zero tail: peeks near the end stay in range)
        intensity = hd["mode"] == 1 and bool(hd["mode_ext"] & 1)
        ms = hd["mode"] == 1 and bool(hd["mode_ext"] & 2)
        sf0 = [None] * nch
        for gr in range(ngr):
            xr = np.zeros((nch, 576))
            for ch in range(nch):
                g = si["gr"][gr][ch]
                start = br.pos
                if g["block_type"] == 2 and g["mixed"] and row == 8:
                    raise Mp3Error("mixed blocks at 8 kHz (MPEG-2.5) are not built")
                if hd["lsf"]:
                    long_sf, short_sf = _scalefactors_lsf(br, g, intensity and ch == 1)
                else:
                    long_sf, short_sf = _scalefactors(br, g, si["scfsi"][ch], sf0[ch] if gr == 1 else None)
                if gr == 0:
                    sf0[ch] = long_sf
                isamp = _huffman(br, g, start + g["part2_3_length"], row, t)
                xr[ch] = _requantise(isamp, g, long_sf, short_sf, row, t)
            if intensity:
                _joint_stereo(xr, np.asarray(isamp) != 0, si["gr"][gr][1], long_sf, short_sf, row, t, ms, hd["lsf"])
            elif ms:
                m, s = xr[0].copy(), xr[1].copy()
                xr[0], xr[1] = (m + s) / np.sqrt(2.0), (m - s) / np.sqrt(2.0)
            for ch in range(nch):
                g = si["gr"][gr][ch]
                lines[ch].append(xr[ch])
                kinds[ch].append(g["block_type"] if g["block_type"] != 2 else (4 if g["mixed"] else 2))
        pos += flen
        frames += 1
    if not frames:
        raise Mp3Error("no decodable frame")
    win = t["window"]
    spf = 576 if info["lsf"] else 1152
    row = RATES.index(rate)
    pcm = np.empty((nch, frames * spf), dtype=np.float32)
    for ch in range(nch):
        # ---- hybrid filter bank (2.4.3.4.9 / .10): reorder, alias reduction, IMDCT, overlap-add, frequency inversion -- over
        # blocks of granules (one numpy call per step and block instead of per granule; blocks small enough to stay in cache)
        tail = np.zeros((32, 18))                                           # second IMDCT half of the granule before
        hist = np.zeros((16, 64))                                           # the last 16 matrixed slots of the block before
        at = 0
        for g0 in range(0, len(lines[ch]), HYBRID_BLOCK):
            s, tail = _hybrid(np.stack(lines[ch][g0:g0 + HYBRID_BLOCK]), np.asarray(kinds[ch][g0:g0 + HYBRID_BLOCK]), tail,
                              row, t)                                       # [slots][32 subbands]
            # ---- polyphase synthesis of the block's slots: V = S N, out[t][j] = sum_i D[64 i + j] V[t - 2 i][j] + D[64 i + 32 + j] V[t - 2 i - 1][32 + j]
            v = np.concatenate([hist, s @ t["synth"]], axis=0)
            hist = v[-16:]
            n = s.shape[0]
            out = np.zeros((n, 32))
            for i in range(8):
                out += win[64 * i:64 * i + 32] * v[16 - 2 * i:16 - 2 * i + n, :32]
                out += win[64 * i + 32:64 * i + 64] * v[15 - 2 * i:15 - 2 * i + n, 32:]
            pcm[ch, at:at + 32 * n] = out.reshape(-1)
            at += 32 * n
    if trim_gapless and info["xing"] and (info["start_pad"] or info["end_pad"]):
        # what FFmpeg's demuxer does with the LAME fields: skip start_pad + 528 + 1 samples, end at
        # frames * 1152 - end_pad + 528 + 1 (the decoder itself delays the signal by 528 + 1 samples)
        total = (info["frames"] if info["frames"] else frames) * spf
        a = info["start_pad"] + 529
        b = min(pcm.shape[1], max(a, total - info["end_pad"] + 529))
        pcm = pcm[:, a:b]
    if clip:
        pcm = np.clip(pcm, -1.0, 1.0)
    return pcm, rate
