"""Host side of the Winograd-domain fp32 conv (csrc/conv1d_wino.h): the F(4, 3) matrices satisfy the
minimal-filtering identity, the weight packer's sub-record order is what the kernel's matrix waves index, and the
kernel's whole index algebra -- channel-pair interleaved raw LDS rows, the per-tile input window, V[k-step][tile][k-row][6], A / B fragment lanes, the
accumulator layout and the 16-byte stores -- emulated lane by lane reproduces a float64 conv.  No GPU needed to get an
offset wrong.  reference: openvoice/modules.py:296-309."""
import numpy as np
import pytest
import torch

from openvoice_amd import _lib, wino

NT, RW = 32, 144


def geo(K):
    G = (K + 2) // 3
    pad = (K - 1) // 2 + wino.LEAD_TAPS[K]      # 'same' padding of the zero-extended filter (csrc/conv1d_wino.h Geo<K>)
    off0 = 8 - pad
    wstart = off0 // 4 * 4
    wlen = off0 - wstart + 3 * (G - 1) + 6
    return G, pad, off0, wstart, (wlen + 3) // 4


def test_f43_matrices_are_a_minimal_filtering_algorithm():
    bt, g, at = (np.array(m, dtype=np.float64) for m in (wino.BT, wino.G, wino.AT))
    rng = np.random.default_rng(0)
    for _ in range(10):
        d, w = rng.standard_normal(6), rng.standard_normal(3)
        y = at @ ((g @ w) * (bt @ d))
        ref = np.array([sum(w[k] * d[i + k] for k in range(3)) for i in range(4)])
        assert np.abs(y - ref).max() < 1e-13


def pack(w):
    lib = _lib.load()
    cout, cin, k = w.shape
    n = lib.ov_conv1d_wino_pack_size(cout, cin, k)
    dst = torch.empty(n, dtype=torch.float32)
    assert lib.ov_conv1d_wino_pack_f32(w.contiguous().data_ptr(), cout, cin, k, dst.data_ptr()) == 0
    return dst.numpy()


@pytest.mark.parametrize("cout,cin,k", [(128, 32, 3), (128, 16, 7), (256, 24, 11), (64, 64, 11), (64, 16, 3), (32, 32, 11), (96, 6, 11)])
def test_weight_packer_sub_record_order(cout, cin, k):
    lib = _lib.load()
    ci = lib.ov_conv1d_wino_chunk(k, cout)
    G = (k + 2) // 3
    w = torch.randn(cout, cin, k, generator=torch.Generator().manual_seed(2)) * (cin * k) ** -0.5
    packed = pack(w)
    npair = ci * G // 4
    assert packed.size == ((cout // 32) * (cin // ci) * npair * 3 + 3) * 256
    assert (packed[-768:] == 0).all()
    lead = wino.LEAD_TAPS[k]                     # zero taps in front of the real ones (K = 7: one at each end)
    wp = np.zeros((cout, cin, 3 * G))
    wp[:, :, lead:lead + k] = w.numpy()
    u = np.einsum("pk,ocgk->ocgp", np.array(wino.G), wp.reshape(cout, cin, G, 3)).astype(np.float32)   # [co][ci][g][p]
    rec = packed[:-768].reshape(cout // 32, cin // ci, npair, 3, 64, 4)
    rng = np.random.default_rng(3)
    for _ in range(2000):
        mt, c, sp, lane, e = (rng.integers(cout // 32), rng.integers(cin // ci), rng.integers(npair), rng.integers(64),
                              rng.integers(12))
        kk = 2 * (2 * sp + e // 6) + (lane >> 5)
        want = u[32 * mt + (lane & 31), c * ci + kk % ci, kk // ci, e % 6]
        # (float64 sums of three products rounded once: equal up to the summation order of the float64 terms)
        assert abs(float(rec[mt, c, sp, e // 4, lane, e % 4]) - float(want)) <= 2.0 ** -23 * abs(float(want))
    # the products the kernel never issues are exact zeros in the stream: point 0 of a group that starts with a zero tap,
    # point infinity of one that ends with one
    zero = [(g, pt) for g in range(G) for pt in (0, 5)
            if (pt == 0 and g == 0 and lead) or (pt == 5 and g == G - 1 and 3 * G - k - lead)]
    assert len(zero) == 6 * G - wino.PRODUCTS_PER_TILE[k]
    for g, pt in zero:
        assert (u[:, :, g, pt] == 0).all()
    assert lib.ov_conv1d_wino_pack_size(100, 128, 3) == 0 and lib.ov_conv1d_wino_pack_size(128, 128, 5) == 0
    assert lib.ov_conv1d_wino_supported(128, 128, 11, 1) == 1 and lib.ov_conv1d_wino_supported(128, 128, 11, 2) == 0
    assert lib.ov_conv1d_wino_supported(128, 128, 11, 5) == 1 and lib.ov_conv1d_wino_supported(256, 256, 3, 3) == 1
    assert lib.ov_conv1d_wino_supported(64, 64, 7, 1) == 1 and lib.ov_conv1d_wino_supported(32, 32, 7, 1) == 0
    assert lib.ov_conv1d_wino_supported(32, 32, 11, 1) == 1 and lib.ov_conv1d_wino_supported(32, 32, 11, 5) == 1


@pytest.mark.parametrize("k,cin,L,cout", [(11, 16, 260, 128), (7, 8, 128, 128), (3, 32, 132, 128), (11, 8, 520, 64), (3, 16, 260, 64),
                                          (11, 4, 260, 32)])
def test_kernel_index_algebra_reproduces_the_conv(k, cin, L, cout):
    """One M-block (128 rows), the helper / matrix wave arithmetic of conv1d_wino_kernel replayed with numpy indexing
    exactly as the kernel forms its addresses (fp32 transforms, float64 accumulation so that only indices are on trial)."""
    lib = _lib.load()
    ci_chunk = lib.ov_conv1d_wino_chunk(k, cout)
    G, pad, off0, wstart, nb128 = geo(k)
    slope = 0.1
    nmt = cout // 32          # (a 64-row layer: two row fragments x two column sub-blocks per workgroup -- at dilation 1
    gen = torch.Generator().manual_seed(5)   # the sub-blocks are consecutive 128-column ranges, so the same replay holds)
    w = torch.randn(cout, cin, k, generator=gen) * (cin * k) ** -0.5
    x = torch.randn(cin, L, generator=gen).numpy()
    bias = torch.randn(cout, generator=gen).numpy()
    packed = pack(w)
    nchunks, kr = cin // ci_chunk, ci_chunk * G
    npair = kr // 4
    ntiles = (L + 127) // 128
    out = np.zeros((cout, L))
    bt, at = np.array(wino.BT, dtype=np.float32), np.array(wino.AT, dtype=np.float64)
    for tile in range(ntiles):
        t0 = tile * 128
        y = np.zeros((nmt, 6, 32, 32))                    # [row fragment][p][row in fragment][tile n]
        y[:, 1] = bias.reshape(nmt, 32)[:, :, None]
        for c in range(nchunks):
            # LDS images exactly as the kernel lays them out: raw[pair][raw index][channel of the pair] (item idx = one
            # 4-column vector of BOTH channels at 8 idx floats), V[k-step = g * CI / 2 + pair][tile][channel of the pair][6]
            npr = ci_chunk // 2
            raw = np.zeros(ci_chunk * RW, dtype=np.float32)
            for idx in range(npr * RW // 4):              # staging items
                pr, c4 = divmod(idx, RW // 4)
                t = t0 - 8 + 4 * c4
                for e in range(4):
                    for ch in range(2):
                        if 0 <= t + e < L:
                            v = x[c * ci_chunk + 2 * pr + ch, t + e]
                            raw[8 * idx + 2 * e + ch] = v if v > 0 else v * slope
            V = np.zeros(kr * NT * 6, dtype=np.float32)
            for idx in range(npr * NT):                   # transform items
                tl, pr = idx & 31, idx >> 5
                src = 2 * (pr * RW + wstart + 4 * tl)     # float offset of the window (kernel: tsrc / 4)
                for ch in range(2):
                    win = raw[src + ch: src + ch + 8 * nb128: 2]
                    for g in range(G):
                        o = off0 - wstart + 3 * g
                        dst = 12 * ((g * npr + pr) * NT + tl) + 6 * ch
                        V[dst:dst + 6] = bt @ win[o:o + 6]
                        if g == 0 and wino.LEAD_TAPS[k]:
                            V[dst] = np.nan                # the helpers do not store the points of products never issued
                        if g == G - 1 and 3 * G - k - wino.LEAD_TAPS[k]:
                            V[dst + 5] = np.nan
            for wave in range(nmt):                        # matrix waves: sub-records -> A fragments, V -> B fragments
                mt = wave
                base = (mt * nchunks + c) * npair * 3
                for s in range(kr // 2):
                    sp, s2 = s >> 1, s & 1
                    for q in range(6):
                        e = s2 * 6 + q
                        g = 2 * s // ci_chunk
                        if (q == 0 and g == 0 and wino.LEAD_TAPS[k]) or (q == 5 and g == G - 1 and 3 * G - k - wino.LEAD_TAPS[k]):
                            continue                       # identically zero: never issued (and its V slot never written)
                        sub = packed[(base + sp * 3 + (e >> 2)) * 256:(base + sp * 3 + (e >> 2) + 1) * 256].reshape(64, 4)
                        a = sub[:, e & 3]                  # lane -> A[row = lane & 31][k = lane >> 5]
                        for half in range(2):              # lane (half, n) -> B[k = half][n] at (s NT + n) 12 + 6 half + q
                            b = V[12 * s * NT + 6 * half + q: 12 * (s + 1) * NT: 12]
                            y[wave, q] += np.outer(a[32 * half:32 * half + 32].astype(np.float64), b.astype(np.float64))
        o = np.einsum("ip,wprn->wrni", at, y)              # [wave][row][tile n][i]
        for wave in range(nmt):
            for n in range(32):
                col = t0 + 4 * n
                if col < L:
                    out[32 * wave:32 * wave + 32, col:col + 4] = o[wave, :, n, :]
    xa = np.where(x > 0, x, slope * x)
    cpad = (k - 1) // 2                                    # the conv's own 'same' padding (`pad` counts the leading zero taps too)
    xp = np.pad(xa, ((0, 0), (cpad, cpad)))
    ref = bias[:, None] + sum(w.numpy()[:, :, j].astype(np.float64) @ xp[:, j:j + L] for j in range(k))
    assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()


@pytest.mark.parametrize("k,dil,cin,L,cout", [(11, 3, 8, 300, 128), (11, 5, 8, 244, 128), (7, 5, 8, 100, 128), (3, 3, 16, 256, 128),
                                              (11, 3, 4, 600, 64), (7, 5, 8, 488, 64), (11, 5, 4, 1000, 32), (11, 3, 2, 2100, 32)])
def test_dilated_kernel_index_algebra_reproduces_the_conv(k, dil, cin, L, cout):
    """Dilated instances (two fragments per wave): tile n = rc * J + jt holds the outputs rc + dil (4 jt + i) of a
    4 J dil-column block; raw rows start PADA columns before the block; a tile reads 3 (G - 1) + 6 inputs dil apart; the
    outputs leave through an 8-row x 256-column stage and are stored 16 bytes per lane.  Replayed with the kernel's
    own index formulas."""
    lib = _lib.load()
    ci_chunk = lib.ov_conv1d_wino_chunk(k, cout)
    G = (k + 2) // 3
    NTS = 64                       # tiles per sub-block (one matrix wave's two fragments)
    NB = {32: 4, 64: 2}.get(cout, 1)   # sub-blocks per workgroup: 2 on a 64-row layer, 4 on a 32-row layer
    NT = NTS * NB
    J = NTS // dil
    ncols = 4 * J * dil            # columns per sub-block
    ncol = ncols * NB
    padd = ((k - 1) // 2 + wino.LEAD_TAPS[k]) * dil      # 'same' padding of the zero-extended filter (GeoD::PADD)
    pada = (padd + 3) // 4 * 4
    nv = 3 * (G - 1) + 6
    rw = (pada - padd + dil - 1 + dil * (4 * (J - 1) + nv - 1) + 1 + 3) // 4 * 4 + (NB - 1) * ncols
    slope = 0.1
    nmt = cout // 32
    gen = torch.Generator().manual_seed(7)
    w = torch.randn(cout, cin, k, generator=gen) * (cin * k) ** -0.5
    x = torch.randn(cin, L, generator=gen).numpy()
    bias = torch.randn(cout, generator=gen).numpy()
    packed = pack(w)
    nchunks, kr = cin // ci_chunk, ci_chunk * G
    npair = kr // 4
    ntiles = (L + ncol - 1) // ncol
    out = np.full((cout, L), np.nan)
    bt, at = np.array(wino.BT, dtype=np.float32), np.array(wino.AT, dtype=np.float64)
    for tile in range(ntiles):
        t0 = tile * ncol
        y = np.zeros((nmt, 6, 32, NT))
        y[:, 1] = bias.reshape(nmt, 32)[:, :, None]
        for c in range(nchunks):
            npr = ci_chunk // 2
            raw = np.zeros(ci_chunk * rw, dtype=np.float32)   # raw[pair][raw index][channel of the pair]
            for idx in range(npr * rw // 4):
                pr, c4 = divmod(idx, rw // 4)
                t = t0 - pada + 4 * c4
                for e in range(4):
                    for ch in range(2):
                        if 0 <= t + e < L:
                            v = x[c * ci_chunk + 2 * pr + ch, t + e]
                            raw[8 * idx + 2 * e + ch] = v if v > 0 else v * slope
            V = np.zeros(kr * NT * 6, dtype=np.float32)       # V[k-step][tile][channel of the pair][6]
            for idx in range(npr * NT):
                tile, pr = idx & (NT - 1), idx // NT
                sub, tl = divmod(tile, NTS)
                rc0 = tl // J
                rc, jt = (rc0, tl - rc0 * J) if rc0 < dil else (0, 0)
                s0 = 2 * (pr * rw + (pada - padd) + sub * ncols + rc + 4 * dil * jt)
                for ch in range(2):
                    win = raw[s0 + ch: s0 + ch + 2 * dil * nv: 2 * dil]
                    assert win.size == nv
                    for g in range(G):
                        dst = 12 * ((g * npr + pr) * NT + tile) + 6 * ch
                        V[dst:dst + 6] = bt @ win[3 * g: 3 * g + 6]
                        if g == 0 and wino.LEAD_TAPS[k]:
                            V[dst] = np.nan                # (points of products never issued are not stored)
                        if g == G - 1 and 3 * G - k - wino.LEAD_TAPS[k]:
                            V[dst + 5] = np.nan
            for wave in range(nmt):
                base = (wave * nchunks + c) * npair * 3
                for s in range(kr // 2):
                    sp, s2 = s >> 1, s & 1
                    for q in range(6):
                        e = s2 * 6 + q
                        g = 2 * s // ci_chunk
                        if (q == 0 and g == 0 and wino.LEAD_TAPS[k]) or (q == 5 and g == G - 1 and 3 * G - k - wino.LEAD_TAPS[k]):
                            continue                       # identically zero: never issued (and its V slot never written)
                        sub = packed[(base + sp * 3 + (e >> 2)) * 256:(base + sp * 3 + (e >> 2) + 1) * 256].reshape(64, 4)
                        a = sub[:, e & 3]
                        for half in range(2):
                            b = V[12 * s * NT + 6 * half + q: 12 * (s + 1) * NT: 12]
                            y[wave, q] += np.outer(a[32 * half:32 * half + 32].astype(np.float64), b.astype(np.float64))
        o = np.einsum("ip,wprn->wrni", at, y)              # [row fragment][row][tile][i]
        for sub in range(NB):                              # each matrix wave stages its own sub-block
            stage = np.full((nmt, 32, 256), np.nan)        # (the kernel walks it 8 rows at a time)
            for tl in range(NTS):
                rc0 = tl // J
                if rc0 >= dil:
                    continue
                cb = rc0 + 4 * dil * (tl - rc0 * J)
                for i in range(4):
                    stage[:, :, cb + i * dil] = o[:, :, sub * NTS + tl, i]
            for lane in range(64):
                col = t0 + sub * ncols + 4 * lane
                if 4 * lane < ncols and col < L:
                    out[:, col:col + 4] = stage[:, :, 4 * lane:4 * lane + 4].reshape(cout, 4)
    xa = np.where(x > 0, x, slope * x)
    cpad = (k - 1) // 2 * dil
    xp = np.pad(xa, ((0, 0), (cpad, cpad)))
    ref = bias[:, None] + sum(w.numpy()[:, :, j].astype(np.float64) @ xp[:, j * dil:j * dil + L] for j in range(k))
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()
