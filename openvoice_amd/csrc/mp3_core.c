/* Host-side helper of openvoice_amd/mp3.py (the MP3 input of the audio boundary; reference call sites openvoice/api.py:123,144
 * through librosa -> audioread -> FFmpeg): the Huffman decoding of one (granule, channel) -- big-values pairs over three
 * regions, then the count1 quadruples -- which is the one per-sample loop of the decoder that numpy cannot express
 * (ISO/IEC 11172-3, 2.4.2.7 "huffmancodebits", Annex B Table B.7).  Plain C, no tables of its own: the look-up tables
 * are built by mp3.py from mp3_tables.npz and passed in.  mp3.py decodes the same way in Python when this library is
 * not built; tests/test_mp3_cpu.py holds both against the FFmpeg golden vectors and against each other.
 *   gcc -O2 -fPIC -shared -o ../libov_mp3.so mp3_core.c */
#include <stdint.h>

static inline uint32_t peek(const uint8_t* buf, int64_t nbytes, int64_t pos, int n) {
  /* n <= 24 bits starting at bit `pos` (MSB first); bytes beyond the buffer read as zero */
  const int64_t b = pos >> 3;
  uint64_t w = 0;
  for (int i = 0; i < 5; ++i) w = (w << 8) | (b + i < nbytes ? buf[b + i] : 0u);
  return (uint32_t)((w >> (40 - (pos & 7) - n)) & ((1u << n) - 1u));
}

/* lut entries: length << 8 | x << 4 | y (big values), length << 8 | v (count1); indexed by the next `maxlen` bits.
 * bounds[3]: first line after each region (already limited to 2 * big_values); table id 0 = all-zero region.
 * Returns the bit position after the granule's last whole code word (the caller continues at `end`). */
int64_t ovmp3_huffman(const uint8_t* buf, int64_t nbytes, int64_t pos, int64_t end, const int32_t* bounds,
                      const uint32_t* const* lut, const int32_t* maxlen, const int32_t* linbits,
                      const uint32_t* quad_lut, int32_t quad_maxlen, int32_t* out /* [576 + 2] */) {
  int i = 0;
  for (int k = 0; k < 578; ++k) out[k] = 0;
  for (int region = 0; region < 3; ++region) {
    const int stop = bounds[region];
    if (!lut[region]) {              /* table 0: all zeros, no bits */
      if (i < stop) i = stop;
      continue;
    }
    const uint32_t* t = lut[region];
    const int ml = maxlen[region], lb = linbits[region];
    while (i < stop) {
      const uint32_t e = t[peek(buf, nbytes, pos, ml)];
      pos += e >> 8;
      int x = (e >> 4) & 15, y = e & 15;
      if (lb && x == 15) { x += (int)peek(buf, nbytes, pos, lb); pos += lb; }
      if (x) { if (peek(buf, nbytes, pos, 1)) x = -x; pos += 1; }
      if (lb && y == 15) { y += (int)peek(buf, nbytes, pos, lb); pos += lb; }
      if (y) { if (peek(buf, nbytes, pos, 1)) y = -y; pos += 1; }
      out[i] = x;
      out[i + 1] = y;
      i += 2;
    }
  }
  /* count1 region: quadruples of magnitude <= 1 until the granule's bits run out */
  while (pos < end && i <= 572) {
    const uint32_t e = quad_lut[peek(buf, nbytes, pos, quad_maxlen)];
    pos += e >> 8;
    const int v = e & 15;
    int vals[4] = {0, 0, 0, 0};
    for (int n = 0; n < 4; ++n)
      if (v & (8 >> n)) { vals[n] = peek(buf, nbytes, pos, 1) ? -1 : 1; pos += 1; }
    if (pos > end) break;            /* ran past the granule: the last quadruple is stuffing, not data */
    for (int n = 0; n < 4; ++n) out[i + n] = vals[n];
    i += 4;
  }
  return pos;
}

int ovmp3_version(void) { return 1; }
