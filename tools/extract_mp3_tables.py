#!/usr/bin/env python3
"""Normative tables of ISO/IEC 11172-3 Layer III for openvoice_amd/mp3.py -> openvoice_amd/mp3_tables.npz.

A Layer III decoder needs the standard's Huffman code tables (Annex B, Table B.7: 15 big-value tables + the two count1
tables) and its synthesis window (Table B.3, D[i] = 32 C[i]); neither follows from a formula, and this image has no
network and no copy of the standard.  It does have a bundled Chromium (the `kaleido` package's executable) whose FFmpeg
build carries the same tables as constant data.  This script -- run ONCE in the build container, output committed --
locates them by content (the first entries of Tables B.7-2 and B.3 are known), reads them out and validates each one
structurally before anything is written:

  * every Huffman table is a COMPLETE PREFIX CODE: Kraft sum of 2^-len == 1 exactly and no codeword is a prefix of
    another (a single wrong length or code breaks it);
  * the window has the standard's structure (D[0] = 0, |D| peaks at the centre tap 1.144989014 = 75038 / 65536, the 64-tap
    sub-blocks alternate in sign as the polyphase prototype does);
  * the scalefactor-band widths found next to them sum to 576 (long) / 192 (short) per sampling rate.

openvoice_amd/mp3.py is then checked end to end against that same Chromium's decoder output
(oracle/make_mp3_golden.py -> tests/golden/mp3_*.npz).  Measurement / build tool; nothing at run time reads Chromium."""
import mmap
import os
import struct
import sys

import numpy as np

KALEIDO = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/kaleido"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openvoice_amd", "mp3_tables.npz")
# (table id, x size) in the order the constant arrays are laid out (id 4 and 14 do not exist; 16..23 and 24..31 share one
# table each and differ in linbits)
TABLES = [(1, 2), (2, 3), (3, 3), (5, 4), (6, 4), (7, 6), (8, 6), (9, 6), (10, 8), (11, 8), (12, 8), (13, 16), (15, 16),
          (16, 16), (24, 16)]


def kraft_and_prefix(codes, lens):
    total = sum(2.0 ** -int(n) for n in lens)
    assert total == 1.0, f"Kraft sum {total}"
    words = sorted((format(int(c), "b").zfill(int(n)) for c, n in zip(codes, lens)))
    assert all(int(c) < (1 << int(n)) for c, n in zip(codes, lens)), "code wider than its length"
    for a, b in zip(words, words[1:]):
        assert not b.startswith(a), f"{a} is a prefix of {b}"


def main():
    with open(KALEIDO, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    out = {}
    # ---- Huffman tables: lens (uint8) then codes (uint16) per table in TABLES order; constant arrays of 16 bytes or more
    # are 16-byte aligned, smaller ones naturally (what the Kraft check below confirms table by table)
    pos = mm.find(bytes([1, 3, 2, 3]) + struct.pack("<4H", 1, 1, 1, 0) + bytes([1, 3, 6, 3, 3, 5, 5, 5, 6]))
    assert pos > 0, "Table B.7-1 not found"
    align = lambda p, nbytes, elem: (p + 15) // 16 * 16 if nbytes >= 16 else (p + elem - 1) // elem * elem
    for tid, xs in TABLES:
        n = xs * xs
        pos = align(pos, n, 1)
        lens = np.frombuffer(mm[pos:pos + n], dtype=np.uint8).copy()
        pos = align(pos + n, 2 * n, 2)
        codes = np.frombuffer(mm[pos:pos + 2 * n], dtype="<u2").copy()
        pos += 2 * n
        kraft_and_prefix(codes, lens)
        out[f"huff{tid}_len"], out[f"huff{tid}_code"] = lens.reshape(xs, xs), codes.reshape(xs, xs)
        print(f"Table B.7-{tid}: {xs} x {xs}, max length {lens.max()}: complete prefix code")
    # ---- count1 tables A / B: lens[2][16], codes[2][16] (uint8), just before the scalefactor-band widths
    bsl = bytes([4, 4, 4, 4, 4, 4, 6, 6, 8, 8, 10, 12, 16, 20, 24, 28, 34, 42, 50, 54, 76, 158])
    sfb = mm.find(bsl)
    assert sfb > 64
    quad = np.frombuffer(mm[sfb - 64:sfb], dtype=np.uint8).copy().reshape(2, 2, 16)     # [lens | codes][A | B][16]
    for t in range(2):
        kraft_and_prefix(quad[1, t], quad[0, t])
    out["count1_len"], out["count1_code"] = quad[0], quad[1]
    print("count1 tables A, B: complete prefix codes")
    # ---- scalefactor-band widths: long[9][22], short[9][13] (uint8), one row per sampling rate
    long_w = np.frombuffer(mm[sfb:sfb + 9 * 22], dtype=np.uint8).copy().reshape(9, 22)
    sfs = mm.find(bytes([4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56]))       # short-block widths at 44.1 kHz
    assert sfs > 0, "short scalefactor-band widths not found"
    short_w = np.frombuffer(mm[sfs:sfs + 9 * 13], dtype=np.uint8).copy().reshape(9, 13)
    assert (long_w.sum(1) == 576).all() and (short_w.sum(1) == 192).all(), (long_w.sum(1), short_w.sum(1))
    # rows: 44100, 48000, 32000 (MPEG-1), 22050, 24000, 16000 (MPEG-2), 11025, 12000, 8000 Hz (MPEG-2.5)
    out["sfb_long_width"], out["sfb_short_width"] = long_w, short_w
    print("scalefactor bands: long", long_w.tolist(), "short", short_w.tolist())
    # ---- synthesis window: first half (257 int32) of C[i] * 2^21 = D[i] * 65536
    w = mm.find(struct.pack("<16i", 0, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2, -3, -3, -4, -4, -5))
    assert w > 0, "Table B.3 not found"
    half = np.frombuffer(mm[w:w + 257 * 4], dtype="<i4").copy()
    assert half[0] == 0 and half[256] == 75038 and np.abs(half).max() == 75038, half[-4:]
    out["window_half"] = half
    print("synthesis window: 257 entries, centre tap", half[256], "/ 65536 =", half[256] / 65536.0)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
