#!/usr/bin/env python3
"""Synthetic-bitstream golden vectors for openvoice_amd/mp3.py: the MPEG-2 / 2.5 (LSF) half, and the corners of MPEG-1 the
five real files do not reach (mixed blocks, scfsi patterns, the preflag bit, a 511-byte-deep bit reservoir, 32 / 48 kHz).

No LSF stream exists in this image (the reference's resources are MPEG-1), and nothing here can encode audio to MP3.  What
a decoder test needs, though, is not audio but VALID BITSTREAMS with known decodes: this script writes Layer III frames
directly -- random quantised spectra, scale factors, block types (long / start / short / stop / mixed), table selections,
region splits, MS stereo -- with its own bit packer (header, MPEG-1 or LSF side information, their scale-factor codings,
Huffman coding of the big-values and count1 regions from the standard's tables; for MPEG-1 the main data of all frames laid
out as one continuous stream so that granules start in earlier frames), and has the image's Chromium (FFmpeg) decode them
through oracle/make_mp3_golden.chromium_decode.  The streams and their decodes are committed as tests/golden/mp3_lsf_*.npz /
mp3_syn_*.npz; tests/test_mp3_cpu.py decodes the stream with openvoice_amd.mp3 and compares.
Test infrastructure, build container only.

    python oracle/make_mp3_synthetic_golden.py"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from openvoice_amd import mp3  # noqa: E402

LINBITS = mp3.LINBITS
TABLE_OF = mp3.TABLE_OF
TABLE_MAX = {0: 0, 1: 1, 2: 2, 3: 2, 5: 3, 6: 3, 7: 5, 8: 5, 9: 5, 10: 7, 11: 7, 12: 7, 13: 15, 15: 15}


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def __len__(self):
        return len(self.bits)

    def tobytes(self, nbytes):
        b = self.bits + [0] * (nbytes * 8 - len(self.bits))
        assert len(b) == nbytes * 8, (len(self.bits), nbytes * 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def select_for(maxval, rng):
    """A table_select whose code table can carry values up to ``maxval`` (random among the valid ones)."""
    ok = []
    for sel in range(32):
        if sel in (4, 14):
            continue
        tid = TABLE_OF[sel]
        cap = (15 + (1 << LINBITS[sel]) - 1) if sel >= 16 else TABLE_MAX[tid]
        if (maxval == 0 and sel == 0) or (maxval > 0 and sel != 0 and cap >= maxval):
            ok.append(sel)
    return ok[int(rng.integers(len(ok)))] if maxval > 0 else 0


def encode_granule(rng, raw, row, force_kind=None, force=None):
    force = force or {}
    """One (granule, channel): random content -> (side-info fields, main-data BitWriter)."""
    li = np.concatenate([[0], np.cumsum(raw["sfb_long_width"][row])]).astype(int)
    si = np.concatenate([[0], np.cumsum(raw["sfb_short_width"][row])]).astype(int)
    kind = force_kind if force_kind is not None else int(rng.integers(0, 6))   # 0,1 long; 2 start; 3 short; 4 stop; 5 mixed
    g = dict(window_switching=int(kind >= 2))
    g["block_type"] = {0: 0, 1: 0, 2: 1, 3: 2, 4: 3, 5: 2}[kind]
    g["mixed"] = int(kind == 5)
    # ---- quantised spectrum: big-values pairs decaying with frequency, then +-1 quadruples, then zeros
    nbig = int(rng.integers(*force.get("nbig", (20, 140))))              # pairs
    ncount1 = int(rng.integers(0, 30))                  # quadruples
    isamp = np.zeros(576, dtype=np.int64)
    env = np.maximum(force.get("floor", 1.0), 40.0 * np.exp(-np.arange(2 * nbig) / (10.0 + 60.0 * rng.random())))
    mags = np.floor(rng.random(2 * nbig) * env).astype(np.int64)
    if rng.random() < force.get("escape_p", 0.3):
        mags[int(rng.integers(0, 8))] = int(rng.integers(16, 400))       # an escape (linbits) value now and then
    isamp[:2 * nbig] = mags * rng.choice([-1, 1], 2 * nbig)
    c1 = rng.integers(0, 2, 4 * ncount1) * rng.choice([-1, 1], 4 * ncount1)
    isamp[2 * nbig:2 * nbig + 4 * ncount1] = c1
    g["big_values"] = nbig
    # ---- regions and tables
    if g["window_switching"]:
        g["region0_count"], g["region1_count"] = (8 if g["block_type"] == 2 and not g["mixed"] else 7), 36
        r1 = 3 * int(si[3]) if g["block_type"] == 2 else int(li[8])
        r2 = 576
        g["subblock_gain"] = [int(v) for v in rng.integers(0, 3, 3)]
    else:
        g["region0_count"], g["region1_count"] = int(rng.integers(0, 16)), int(rng.integers(0, 8))
        r1 = int(li[min(g["region0_count"] + 1, 22)])
        r2 = int(li[min(g["region0_count"] + g["region1_count"] + 2, 22)])
        g["subblock_gain"] = [0, 0, 0]
    big = 2 * nbig
    bounds = (min(r1, big), min(r2, big), big)
    lo = 0
    sels = []
    for hi in bounds:
        seg = np.abs(isamp[lo:hi])
        sels.append(select_for(int(seg.max()) if len(seg) else 0, rng) if hi > lo else int(rng.integers(0, 32) if False else 0))
        lo = max(lo, hi)
    if g["window_switching"]:
        sels[2] = 0
    g["table_select"] = sels
    g["global_gain"] = int(rng.integers(126, 156))       # (the decode stays inside +-1: nothing saturates)
    g["scalefac_scale"], g["count1table_select"] = force.get("sscale", int(rng.integers(0, 2))), int(rng.integers(0, 2))
    # ---- LSF scale factors: pick a compress value of a random range, factors below 2^slen
    rng_range = force.get("range", int(rng.integers(0, 3)))
    if rng_range == 0:
        sl = [int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4))]
        sfc = ((sl[0] * 5 + sl[1]) << 4) | (sl[2] << 2) | sl[3]
        assert sfc < 400
    elif rng_range == 1:
        sl = [int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), 0]
        sfc = 400 + (((sl[0] * 5 + sl[1]) << 2) | sl[2])
        assert sfc < 500
    else:
        sl = [int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0, 0]
        sfc = 500 + sl[0] * 3 + sl[1]
        assert sfc < 512
    g["scalefac_compress"] = sfc
    bk = 0 if g["block_type"] != 2 else (2 if g["mixed"] else 1)
    if force.get("lsf_intensity_right"):
        # the right channel of an LSF intensity-stereo frame: bit 0 = intensity scale, the rest picks one of three other
        # partitions (ISO/IEC 13818-3 2.4.3.2); widths mostly <= 3 bits so that legal positions (< 16) are the rule
        small = lambda hi: int(rng.integers(1, min(hi, 4))) if rng.random() < 0.8 else int(rng.integers(0, hi))
        rng_range = 3 + int(rng.integers(0, 3))
        if rng_range == 3:
            sl = [small(5), small(6), small(6), 0]
            isfc = sl[0] * 36 + sl[1] * 6 + sl[2]
            assert isfc < 180
        elif rng_range == 4:
            sl = [small(4), small(4), small(4), 0]
            isfc = 180 + ((sl[0] << 4) | (sl[1] << 2) | sl[2])
            assert isfc < 244
        else:
            sl = [small(4), int(rng.integers(0, 3)), 0, 0]
            isfc = 244 + sl[0] * 3 + sl[1]
            assert isfc < 256
        g["scalefac_compress"] = (isfc << 1) | int(rng.integers(0, 2))
    if force.get("mpeg1"):
        # MPEG-1 scale factors (ISO 11172-3 2.4.2.7): 4-bit scalefac_compress -> (slen1, slen2); long blocks in four band
        # groups that granule 1 may inherit from granule 0 (scfsi); a preflag BIT
        sfc = int(rng.integers(0, 16))
        g["scalefac_compress"] = sfc
        g["preflag"] = int(rng.integers(0, 2)) if g["block_type"] != 2 else 0
        slen1, slen2 = mp3.SLEN[0][sfc], mp3.SLEN[1][sfc]
        bw = BitWriter()
        rnd = lambda bits: bw.put(int(rng.integers(0, 1 << bits)) if bits else 0, bits)
        if g["block_type"] == 2:
            if g["mixed"]:
                for _ in range(8):
                    rnd(slen1)
                for _ in range(3 * 3):
                    rnd(slen1)
            else:
                for _ in range(6 * 3):
                    rnd(slen1)
            for _ in range(6 * 3):
                rnd(slen2)
        else:
            scfsi = force.get("scfsi", [0, 0, 0, 0])
            for grp, cnt in enumerate((6, 5, 5, 5)):
                if not scfsi[grp]:
                    for _ in range(cnt):
                        rnd(slen1 if grp < 2 else slen2)
        sl = None
    if "sl" in force:
        sl = list(force["sl"])
        sfc = {0: ((sl[0] * 5 + sl[1]) << 4) | (sl[2] << 2) | sl[3], 1: 400 + (((sl[0] * 5 + sl[1]) << 2) | sl[2]), 2: 500 + sl[0] * 3 + sl[1]}[rng_range]
        g["scalefac_compress"] = sfc
    if sl is not None:
        bw = BitWriter()
        for n, bits in zip(mp3.NR_OF_SFB[rng_range][bk], sl):
            for _ in range(n):
                bw.put(int(rng.integers(0, 1 << bits)) if bits else 0, bits)
    # ---- Huffman: big values per region, then count1
    lo = 0
    for region, hi in enumerate(bounds):
        sel = sels[region]
        tid, linbits = TABLE_OF[sel], LINBITS[sel]
        for i in range(lo, hi, 2):
            x, y = int(isamp[i]), int(isamp[i + 1])
            if tid == 0:
                assert x == 0 and y == 0
                continue
            ax, ay = min(abs(x), 15), min(abs(y), 15)
            bw.put(int(raw[f"huff{tid}_code"][ax, ay]), int(raw[f"huff{tid}_len"][ax, ay]))
            if linbits and ax == 15:
                bw.put(abs(x) - 15, linbits)
            if x:
                bw.put(1 if x < 0 else 0, 1)
            if linbits and ay == 15:
                bw.put(abs(y) - 15, linbits)
            if y:
                bw.put(1 if y < 0 else 0, 1)
        lo = max(lo, hi)
    q = g["count1table_select"]
    for i in range(big, big + 4 * ncount1, 4):
        vals = isamp[i:i + 4]
        v = int(sum((1 if vals[n] else 0) << (3 - n) for n in range(4)))
        bw.put(int(raw["count1_code"][q][v]), int(raw["count1_len"][q][v]))
        for n in range(4):
            if vals[n]:
                bw.put(1 if vals[n] < 0 else 0, 1)
    g["part2_3_length"] = len(bw)
    assert len(bw) < 4096
    return g, bw


def make_stream(seed, version_bits, sr_index, stereo, frames=60, force=None, kinds=None, intensity=False):
    raw = np.load(os.path.join(REPO, "openvoice_amd", "mp3_tables.npz"))
    rng = np.random.default_rng(seed)
    fam = {2: 1, 0: 2}[version_bits]
    row = 3 * fam + sr_index
    rate = mp3.RATES[row]
    bri = 14                                             # 160 kbit/s: room for any granule generated above
    flen = 72000 * mp3.BITRATES_LSF[bri] // rate
    nch = 2 if stereo else 1
    out = bytearray()
    first_ms = bool(seed % 2)
    state = [0] * nch                                    # last block kind per channel (0, 1 long; 2 start; 3 short; 4 stop; 5 mixed)
    for f in range(frames):
        # (the joint-stereo / stereo choice is held for the first frames: FFmpeg's container probing drops leading frames when
        # the header changes right at the start of a stream -- a property of its stream detection, not of Layer III decoding)
        ms = stereo and (rng.random() < 0.5 if f >= 6 else first_ms)
        mode, mode_ext = (3, 0) if not stereo else ((1, 2) if ms else (0, 0))
        if intensity:
            # joint stereo with intensity coding (mode_extension bit 0), with and without MS; a plain / MS frame now and then
            mode, mode_ext = [(1, 1), (1, 3), (1, 1), (1, 3), (1, 2), (0, 0)][int(rng.integers(0, 6)) if f >= 6 else int(first_ms)]
        hdr = (0x7FF << 21) | (version_bits << 19) | (1 << 17) | (1 << 16) | (bri << 12) | (sr_index << 10) | (mode << 6) | (mode_ext << 4)
        # block kinds follow the LEGAL window sequence of the standard (long / stop -> long or start; start -> short or mixed;
        # short / mixed -> short, mixed or stop), per channel: decoders are free to assume it -- FFmpeg's short-block
        # overlap drops the previous block's samples 12..17, which are zero after a start or short block and only then
        if kinds is not None:
            ks = [kinds[f % len(kinds)]] * nch
        else:
            ks = []
            for ch in range(nch):
                prev = state[ch]
                if prev in (0, 1, 4):
                    k = int(rng.choice([0, 1, 2, 2]))
                elif prev == 2:
                    k = int(rng.choice([3, 5]))
                else:                                    # a run of short blocks is all-short or all-mixed: the two lowest
                    k = int(rng.choice([prev, 4, 4]))    # subbands of a mixed block are long ones and follow the same rule
                state[ch] = k
                ks.append(k)
        forces = [force] * nch
        if intensity:
            # both channels in the same block type (the standard requires it of intensity-coded frames); the left channel's
            # spectrum reaches far up (it carries the sum signal), the right channel's ends early: the bands above are the
            # intensity region, the right channel's factors there the positions
            ks[1] = ks[0]
            state[1] = state[0]
            forces = [dict(force or {}, nbig=(120, 220), floor=2.5, escape_p=0.0),
                      dict(force or {}, nbig=(10, 100), lsf_intensity_right=bool(mode_ext & 1))]
        grs = [encode_granule(rng, raw, row, ks[ch], forces[ch]) for ch in range(nch)]
        side = BitWriter()
        side.put(0, 8)                                   # main_data_begin: no reservoir
        side.put(0, 1 if nch == 1 else 2)
        for g, _ in grs:
            side.put(g["part2_3_length"], 12)
            side.put(g["big_values"], 9)
            side.put(g["global_gain"], 8)
            side.put(g["scalefac_compress"], 9)
            side.put(g["window_switching"], 1)
            if g["window_switching"]:
                side.put(g["block_type"], 2)
                side.put(g["mixed"], 1)
                side.put(g["table_select"][0], 5)
                side.put(g["table_select"][1], 5)
                for v in g["subblock_gain"]:
                    side.put(v, 3)
            else:
                for v in g["table_select"]:
                    side.put(v, 5)
                side.put(g["region0_count"], 4)
                side.put(g["region1_count"], 3)
            side.put(g["scalefac_scale"], 1)
            side.put(g["count1table_select"], 1)
        side_len = 9 if nch == 1 else 17
        assert len(side) == side_len * 8, (len(side), side_len)
        main = BitWriter()
        for _, bw in grs:
            main.bits += bw.bits
        body = flen - 4 - side_len
        assert len(main) <= body * 8, (len(main), body * 8)
        out += hdr.to_bytes(4, "big") + side.tobytes(side_len) + main.tobytes(body)
    return bytes(out), rate, nch


def make_stream_v1(seed, sr_index, stereo, frames=50, intensity=False):
    """MPEG-1 frames: two granules, scfsi, 4-bit scalefac_compress, preflag bit -- and a REAL bit reservoir: the main data
    of all frames is one continuous bit stream laid into the frames' data areas back to back, so that a frame's granules
    may start in earlier frames (main_data_begin > 0) exactly as encoders do it."""
    raw = np.load(os.path.join(REPO, "openvoice_amd", "mp3_tables.npz"))
    rng = np.random.default_rng(seed)
    rate = mp3.RATES[sr_index]
    nch = 2 if stereo else 1
    bri = 14 if intensity else 13                        # 256 kbit/s (320 where the left channel carries wide spectra)
    side_len = 17 if nch == 1 else 32
    first_ms = bool(seed % 2)
    state = [0] * nch
    frames_side, frames_main, lens = [], [], []
    for f in range(frames):
        ms = stereo and (rng.random() < 0.5 if f >= 6 else first_ms)
        mode, mode_ext = (3, 0) if not stereo else ((1, 2) if ms else (0, 0))
        if intensity:
            mode, mode_ext = [(1, 1), (1, 3), (1, 1), (1, 3), (1, 2), (0, 0)][int(rng.integers(0, 6)) if f >= 6 else int(first_ms)]
        pad = int(rng.integers(0, 2)) if rate == 44100 else 0
        hdr = (0x7FF << 21) | (3 << 19) | (1 << 17) | (1 << 16) | (bri << 12) | (sr_index << 10) | (pad << 9) | (mode << 6) | (mode_ext << 4)
        grs = [[None] * nch for _ in range(2)]
        scfsi = [[0, 0, 0, 0] for _ in range(nch)]
        for ch in range(nch):
            kinds = []
            for gr in range(2):
                if intensity and ch == 1:                # intensity-coded frames: both channels in the same block type
                    kinds = list(left_kinds)
                    state[1] = state[0]
                    break
                prev = state[ch]
                if prev in (0, 1, 4):
                    k = int(rng.choice([0, 1, 2, 2]))
                elif prev == 2:
                    k = int(rng.choice([3, 5]))
                else:
                    k = int(rng.choice([prev, 4, 4]))
                state[ch] = k
                kinds.append(k)
            if kinds[0] not in (3, 5) and kinds[1] not in (3, 5):     # scfsi only between two long-type granules
                scfsi[ch] = [int(v) for v in rng.integers(0, 2, 4)]
            left_kinds = kinds if ch == 0 else left_kinds
            extra = {}
            if intensity:
                extra = dict(nbig=(100, 180), floor=2.5, escape_p=0.0) if ch == 0 else dict(nbig=(10, 90))
            grs[0][ch] = encode_granule(rng, raw, sr_index, kinds[0], dict(mpeg1=True, **extra))
            grs[1][ch] = encode_granule(rng, raw, sr_index, kinds[1], dict(mpeg1=True, scfsi=scfsi[ch], **extra))
        main = BitWriter()
        for gr in range(2):
            for ch in range(nch):
                main.bits += grs[gr][ch][1].bits
        while len(main) % 8:
            main.bits.append(0)
        frames_main.append(main)
        frames_side.append((hdr, pad, scfsi, grs))
        lens.append(144000 * mp3.BITRATES[bri] // rate + pad)
    # lay the main data out: frame f's bytes go as early as possible, at most 511 bytes before its own data area and never
    # before the end of frame f - 1's
    out = bytearray()
    pool = bytearray()                                   # all data areas so far, concatenated
    cursor = 0                                           # first free byte of the pool
    starts = []
    area_start = []
    for f in range(frames):
        area_start.append(len(pool))
        pool += bytes(lens[f] - 4 - side_len)
        mbytes = frames_main[f].tobytes(len(frames_main[f]) // 8)
        begin = max(cursor, area_start[f] - 511)
        assert begin + len(mbytes) <= len(pool), "granules do not fit: lower the spectra or raise the bit rate"
        pool[begin:begin + len(mbytes)] = mbytes
        starts.append(area_start[f] - begin)             # main_data_begin of frame f
        cursor = begin + len(mbytes)
    for f in range(frames):
        hdr, pad, scfsi, grs = frames_side[f]
        side = BitWriter()
        side.put(starts[f], 9)
        side.put(0, 5 if nch == 1 else 3)
        for ch in range(nch):
            for v in scfsi[ch]:
                side.put(v, 1)
        for gr in range(2):
            for ch in range(nch):
                g = grs[gr][ch][0]
                side.put(g["part2_3_length"], 12)
                side.put(g["big_values"], 9)
                side.put(g["global_gain"], 8)
                side.put(g["scalefac_compress"], 4)
                side.put(g["window_switching"], 1)
                if g["window_switching"]:
                    side.put(g["block_type"], 2)
                    side.put(g["mixed"], 1)
                    side.put(g["table_select"][0], 5)
                    side.put(g["table_select"][1], 5)
                    for v in g["subblock_gain"]:
                        side.put(v, 3)
                else:
                    for v in g["table_select"]:
                        side.put(v, 5)
                    side.put(g["region0_count"], 4)
                    side.put(g["region1_count"], 3)
                side.put(g["preflag"], 1)
                side.put(g["scalefac_scale"], 1)
                side.put(g["count1table_select"], 1)
        assert len(side) == side_len * 8
        a = area_start[f]
        out += hdr.to_bytes(4, "big") + side.tobytes(side_len) + bytes(pool[a:a + lens[f] - 4 - side_len])
    return bytes(out), rate, nch, max(starts)


CASES_V1 = [("mpeg1_44100_mono_reservoir", 21, 0, False), ("mpeg1_48000_stereo_ms", 22, 1, True), ("mpeg1_32000_mono", 23, 2, False)]
CASES = [("mpeg2_22050_mono", 11, 2, 0, False), ("mpeg2_24000_stereo_ms", 12, 2, 1, True), ("mpeg2_16000_mono", 13, 2, 2, False),
         ("mpeg25_11025_stereo", 14, 0, 0, True), ("mpeg25_12000_mono", 15, 0, 1, False)]


CASES_INTENSITY = [("mpeg1_44100_intensity", 31, "v1", 0), ("mpeg1_32000_intensity", 32, "v1", 2),
                   ("mpeg2_22050_intensity", 41, 2, 0), ("mpeg2_16000_intensity", 42, 2, 2), ("mpeg25_11025_intensity", 43, 0, 0)]


def main():
    from kaleido.scopes.plotly import PlotlyScope
    from make_mp3_golden import FAKE_PLOTLY, chromium_decode
    js = os.path.join("/tmp", "ov_fake_plotly.js")
    with open(js, "w") as fh:
        fh.write(FAKE_PLOTLY)
    scope = PlotlyScope(plotlyjs=js)
    only = sys.argv[1] if len(sys.argv) > 1 else ""      # e.g. "intensity": regenerate only the files whose name holds it
    for name, seed, vbits, sri in CASES_INTENSITY:
        if only not in name:
            continue
        if vbits == "v1":
            data, rate, nch, _ = make_stream_v1(seed, sri, True, intensity=True)
        else:
            data, rate, nch = make_stream(seed, vbits, sri, True, intensity=True)
        pcm = chromium_decode(data, nch, rate, scope)
        mine, _ = mp3.decode(data)
        n = min(pcm.shape[1], mine.shape[1])
        print(name, "stream", len(data), "bytes; chromium", pcm.shape, "peak", float(np.abs(pcm).max()), "| this decoder",
              mine.shape, "max-abs difference", float(np.abs(pcm[:, :n] - mine[:, :n]).max()))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", f"mp3_syn_{name}.npz"), stream=np.frombuffer(data, dtype=np.uint8),
                            pcm=pcm, rate=rate, channels=nch)
    for name, seed, sri, stereo in CASES_V1:
        if only not in name:
            continue
        data, rate, nch, deepest = make_stream_v1(seed, sri, stereo)
        pcm = chromium_decode(data, nch, rate, scope)
        mine, _ = mp3.decode(data)
        n = min(pcm.shape[1], mine.shape[1])
        print(name, "stream", len(data), "bytes, reservoir up to", deepest, "bytes; chromium", pcm.shape, "peak",
              float(np.abs(pcm).max()), "| this decoder", mine.shape, "max-abs difference", float(np.abs(pcm[:, :n] - mine[:, :n]).max()))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", f"mp3_syn_{name}.npz"), stream=np.frombuffer(data, dtype=np.uint8),
                            pcm=pcm, rate=rate, channels=nch)
    for name, seed, vbits, sri, stereo in CASES:
        if only not in name:
            continue
        data, rate, nch = make_stream(seed, vbits, sri, stereo)
        pcm = chromium_decode(data, nch, rate, scope)
        mine, r2 = mp3.decode(data)
        n = min(pcm.shape[1], mine.shape[1])
        print(name, "stream", len(data), "bytes; chromium", pcm.shape, "peak", float(np.abs(pcm).max()), "| this decoder",
              mine.shape, "max-abs difference", float(np.abs(pcm[:, :n] - mine[:, :n]).max()))
        np.savez_compressed(os.path.join(REPO, "tests", "golden", f"mp3_lsf_{name}.npz"), stream=np.frombuffer(data, dtype=np.uint8),
                            pcm=pcm, rate=rate, channels=nch)


if __name__ == "__main__":
    main()
