"""Host side of the split-precision conv (csrc/conv1d_split3.h): the three-plane bf16 split is
lossless for fp32, the six retained plane products reproduce an fp32 product to ~2^-25, the weight packer's record order
is what the kernel's matrix waves index, and the kernel's LDS address algebra (DMA swizzle -> operand reads, epilogue
cells -> whole-row stores) is a bijection onto the tensors -- emulated here index by index, because the GPU is not needed
to get an XOR wrong.  reference: openvoice/modules.py:296-309."""
import numpy as np
import pytest
import torch

from openvoice_amd import _lib
from openvoice_amd.split3 import split3_reference


def test_three_bf16_planes_are_lossless_for_fp32():
    gen = torch.Generator().manual_seed(0)
    v = torch.randn(1 << 18, generator=gen) * torch.exp(8 * torch.randn(1 << 18, generator=gen))
    # (domain: 2^-100 <= |v| <= bf16 max = 3.39e38, or 0 -- above it hi overflows to inf, below it the lower planes are
    # bf16 denormals; the generator's activations are O(1))
    v = torch.cat([v, torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 2.0 - 2 ** -22, 3.38e38, -3.38e38, 2.0 ** -100,
                                    255.0 / 256.0 + 2 ** -24, 1.00390625, 0.1, 10.0])])
    pl = split3_reference(v)
    back = (pl[0].float() + pl[1].float()) + pl[2].float()
    assert torch.equal(back, v)
    # the planes shrink by >= 2^-8 each (round to nearest), so dropped products are <= 2^-25 of the leading one
    assert (pl[1].float().abs() <= v.abs() * 2.0 ** -8 + 1e-45).all()
    assert (pl[2].float().abs() <= v.abs() * 2.0 ** -16 + 1e-45).all()


def test_six_plane_products_reach_fp32_accuracy():
    """hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid vs the exact product, in float64: the dropped terms
    (mid*lo, lo*mid, lo*lo) are below 2^-24 of the product -- half an fp32 ulp."""
    gen = torch.Generator().manual_seed(1)
    x, w = torch.randn(1 << 16, generator=gen), torch.randn(1 << 16, generator=gen)
    xp, wp = split3_reference(x).double(), split3_reference(w).double()
    six = xp[0] * wp[0] + xp[0] * wp[1] + xp[1] * wp[0] + xp[0] * wp[2] + xp[2] * wp[0] + xp[1] * wp[1]
    exact = x.double() * w.double()
    rel = ((six - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -24, rel
    three = xp[0] * wp[0] + xp[0] * wp[1] + xp[1] * wp[0]
    rel3 = ((three - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    assert rel3 <= 2.0 ** -15, rel3


@pytest.mark.parametrize("cout,cin,k", [(128, 128, 3), (256, 256, 11), (128, 128, 7)])
def test_weight_packer_record_order(cout, cin, k):
    """Record (((ct * Cin/32 + c) * K + tap) * 3 + plane) * 2 + f, lane (l15, g) -> plane of
    W[32 ct + 16 f + l15][32 c + 8 g .. + 8][tap]; the planes of every weight sum to it exactly; trailing zero record."""
    lib = _lib.load()
    w = torch.randn(cout, cin, k, generator=torch.Generator().manual_seed(2)) * (cin * k) ** -0.5
    n = lib.ov_conv1d_split3_pack_size(cout, cin, k)
    assert n == ((cout // 32) * (cin // 32) * k * 6 + 1) * 512
    dst = torch.empty(n, dtype=torch.int16)
    assert lib.ov_conv1d_split3_pack(w.data_ptr(), cout, cin, k, dst.data_ptr()) == 0
    rec = dst[:-512].view(torch.bfloat16).view(cout // 32, cin // 32, k, 3, 2, 4, 16, 8)   # ct, c, tap, plane, f, g, l15, i
    planes = split3_reference(w)                                                            # (3, cout, cin, k)
    want = planes.view(3, cout // 32, 2, 16, cin // 32, 4, 8, k).permute(1, 4, 7, 0, 2, 5, 3, 6)   # ct, c, tap, plane, f, g, l15, i
    assert torch.equal(rec, want.contiguous())
    assert (dst[-512:] == 0).all()
    assert lib.ov_conv1d_split3_pack_size(100, 128, 3) == 0 and lib.ov_conv1d_split3_supported(128, 128, 11, 5) == 1
    assert lib.ov_conv1d_split3_supported(64, 64, 3, 1) == 1 and lib.ov_conv1d_split3_supported(128, 128, 5, 1) == 0
    assert lib.ov_conv1d_split3_supported(32, 32, 3, 1) == 0 and lib.ov_conv1d_split3_supported(128, 256, 3, 1) == 0


def _check_input_addresses(K, DIL, CIN, L, tile, c, COT=128):
    """conv1d_split3.h, input waves' dma_chunk() against the matrix waves' operand reads (xl_tap / oread); COT = 64:
    2 x 2 matrix waves, the upper pair starts at row trow0 = 64 with JW = 4 time fragments."""
    TT, P1 = 128, (K - 1) * DIL // 2
    NBLK = (TT + 2 * P1 + 15) // 16
    PGI = 2 * CIN
    NCT = COT // 32
    JW = 8 // (4 // NCT)
    lds = np.full(NBLK * 512, -2, dtype=np.int64)            # per bf16 element: global element id, -1 = zero record
    tbase = tile * TT - P1
    for blk in range(NBLK):
        for lane in range(64):
            lrow, sp = lane >> 2, lane & 3
            dof = lrow * PGI + 16 * (sp ^ ((lrow >> 2) & 3))
            t = tbase + blk * 16 + lrow
            d = (blk * 1024 + lane * 16) // 2
            lds[d:d + 8] = -1 if (t < 0 or t >= L) else ((tbase + blk * 16) * PGI + 64 * c + dof) // 2 + np.arange(8)
    for wave in range(4):
        trow0 = (wave // NCT) * 16 * JW
        for tap in range(K):
            for lane in range(64):
                l15, g4 = lane & 15, lane >> 4
                row = trow0 + l15 + tap * DIL
                base = row * 64 + 16 * (g4 ^ ((row >> 2) & 3))
                for j in range(JW):
                    t = tile * TT - P1 + trow0 + 16 * j + l15 + tap * DIL
                    want = np.full(8, -1) if (t < 0 or t >= L) else t * CIN + 32 * c + 8 * g4 + np.arange(8)
                    a = (base + j * 1024) // 2
                    assert (lds[a:a + 8] == want).all(), (wave, tap, lane, j)


def _check_output_addresses(COUT, mb, COT=128):
    """conv1d_split3.h, the matrix waves' epilogue cells (ecell) against the output waves' fetch / store (oblk, ophase,
    odof) -- and thereby the input waves' residual DMA, which uses the same block / phase functions."""
    NCT = COT // 32
    JW = 8 // (4 // NCT)
    NR, OP, PGO = (2 if COT == 128 else 1), 2 * COT, 2 * COUT
    OROWS, RPBO, SPRO = 16384 // OP, 1024 // OP, OP // 16
    NE = 2 if COT == 128 else 1
    NQ = 8 // NE
    covered = np.zeros((128, COT), dtype=np.int64)
    for h in range(NR):
        lds = np.full(16384 // 2, -2, dtype=np.int64)
        for wave in range(4):
            ct, trow0 = wave % NCT, (wave // NCT) * 16 * JW
            erow0 = 0 if COT == 128 else trow0
            for lane in range(64):
                l15, g4 = lane & 15, lane >> 4
                eswz = l15 if COT == 128 else (l15 >> 1) & 7
                for f in range(2):
                    ecell = (erow0 + l15) * OP + 16 * ((4 * ct + 2 * f + (g4 >> 1)) ^ eswz) + 8 * (g4 & 1)
                    for jj in range(4):
                        a = (ecell + jj * 16 * OP) // 2
                        assert (lds[a:a + 4] == -2).all()
                        trow = trow0 + 16 * (4 * h + jj) + l15          # acc[f][4 h + jj] of this wave
                        lds[a:a + 4] = trow * COUT + COT * mb + 32 * ct + 16 * f + 4 * g4 + np.arange(4)
        assert (lds >= 0).all()
        out = np.full(OROWS * COUT, -2, dtype=np.int64)
        for ow in range(2):
            for lane in range(64):
                lrowo, spo = lane // SPRO, lane % SPRO
                for q in range(NQ):
                    for e in range(NE):
                        blk = 4 * q + 2 * ow + e if COT == 128 else 2 * q + ow
                        phase = 4 * (2 * ow + e) + lrowo if COT == 128 else 4 * ow + (lrowo >> 1)
                        odof = lrowo * PGO + 16 * (spo ^ phase)
                        s_, d = (blk * 1024 + lane * 16) // 2, (blk * RPBO * PGO + OP * mb + odof) // 2
                        out[d:d + 8] = lds[s_:s_ + 8]
        for r in range(OROWS):
            got = out[r * COUT + COT * mb: r * COUT + COT * mb + COT]
            assert (got == (OROWS * h + r) * COUT + COT * mb + np.arange(COT)).all(), (h, r)
            covered[OROWS * h + r] += 1
    assert (covered == 1).all()


def test_kernel_lds_address_algebra():
    for K, D in [(11, 1), (11, 5), (3, 1), (7, 3)]:
        for CIN, COT in ((128, 128), (256, 128), (64, 64)):
            for L, tile in [(1000, 0), (1000, 3), (1000, 7), (130, 1), (5, 0)]:
                _check_input_addresses(K, D, CIN, L, tile, CIN // 32 - 1, COT)
    for COUT, mb, COT in [(128, 0, 128), (256, 0, 128), (256, 1, 128), (64, 0, 64)]:
        _check_output_addresses(COUT, mb, COT)
