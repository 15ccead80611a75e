"""Host helpers of the Winograd-domain fp32 conv kernel (csrc/conv1d_wino.h): the ResBlock convs of the MFMA-bound
generator stages with 1.6-2x fewer executed multiplies than the direct implicit GEMM.

``PackedConvWino`` holds a conv's weights transformed (float64, once) into the six-point F(4, 3) domain in MFMA
fragment order; ``launch_conv_wino`` runs ``out = (conv1d(lrelu(x)) + bias [+ res] [+ add]) * scale`` on fp32
(B, C, L) tensors -- the LINEAR epilogue of ``ov_conv1d_f32`` without a mask.
reference: openvoice/modules.py:296-309 (ResBlock1.forward), models.py:280-286 (MRF sum / mean)."""
import ctypes

import torch

from . import _lib

# F(4, 3) with interpolation points 0, 1, -1, 2, -2, infinity (the matrices the kernel and the packer apply)
BT = ((4, 0, -5, 0, 1, 0), (0, -4, -4, 1, 1, 0), (0, 4, -4, -1, 1, 0), (0, -2, -1, 2, 1, 0), (0, 2, -1, -2, 1, 0),
      (0, 4, 0, -5, 0, 1))
G = ((1 / 4, 0, 0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6), (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6),
     (0, 0, 1))
AT = ((1, 1, 1, 1, 1, 0), (0, 1, -1, 2, -2, 0), (0, 1, 1, 4, 4, 0), (0, 1, -1, 8, -8, 1))


# Zero taps in front of the K real ones (csrc/conv1d_wino.h wino_lead): K = 7 is transformed as (0 w0 w1)(w2 w3 w4)(w5 w6 0),
# K = 11 as (w0 w1 w2) ... (w9 w10 0).  A zero tap at the start of a group makes its point-0 weight zero, one at the end its
# point-infinity weight: those products are never issued.
LEAD_TAPS = {3: 0, 7: 1, 11: 0}
# multiply-accumulates the kernel EXECUTES per 4 outputs and (co, ci): 6 per group minus the identically-zero products
# (the direct form needs 4 K: 12 / 28 / 44)
PRODUCTS_PER_TILE = {3: 6, 7: 16, 11: 23}


def supported(cin, cout, K, dil):
    return bool(_lib.call("ov_conv1d_wino_supported", cin, cout, K, dil))


class PackedConvWino:
    """One conv layer in Winograd-domain kernel-ready form (``ov_conv1d_wino_pack_f32``) + fp32 bias."""

    def __init__(self, w_dense, bias, device, dil=1):
        w = w_dense.detach().to(torch.float32).cpu().contiguous()
        self.cout, self.cin, self.K = w.shape
        self.dil = dil
        if not supported(self.cin, self.cout, self.K, dil):
            raise _lib.OvError(f"ov_conv1d_wino_f32: no instance for Cin={self.cin} Cout={self.cout} K={self.K} dil={dil}")
        n = _lib.call("ov_conv1d_wino_pack_size", self.cout, self.cin, self.K)
        packed = torch.empty(n, dtype=torch.float32)
        _lib.call("ov_conv1d_wino_pack_f32", w, self.cout, self.cin, self.K, packed)
        self.w = packed.to(device)
        b = torch.zeros(self.cout) if bias is None else bias.detach().float().cpu()
        self.bias = b.contiguous().to(device)


def launch_conv_wino(layer, x, x_bs, out, out_bs, B, L, in_slope=1.0, scale=1.0, res=None, res_bs=0, add=None, add_bs=0,
                     x_ld=0, out_ld=0, nwg=0, dbg=None, frags=0, col_limit=None, col_limit_scale=1, out_slope=1.0):
    """One launch on torch's current stream of ``x``'s device; strides in elements (``*_ld`` 0 = dense rows).
    ``col_limit`` (int32 [B] on the device) x ``col_limit_scale`` = columns of each utterance that matter: column blocks
    beyond are neither computed nor written (length-aware work list, as ``engine.launch_conv``).  ``out_slope`` < 1 stores
    lrelu(result, out_slope) (only without ``res`` / ``add``): the first conv of a ResBlock pair hands its consumer an
    activated tensor, which then stages it with ``in_slope`` = 1 -- the same values, one leaky ReLU per element instead of one
    per element and launch that reads it."""
    if _lib.use_torch_binding():
        _lib.torch_op("conv1d_wino_f32", x, layer.w, layer.bias, out, res, add, dbg, col_limit,
                      [B, layer.cin, layer.cout, L, x_ld, out_ld, layer.K, layer.dil, nwg, x_bs, out_bs, res_bs, add_bs, frags, col_limit_scale],
                      [in_slope, scale, out_slope])
        return
    p = _lib.ConvWinoParams()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w, p.bias, p.out, p.res, p.add = vp(x), vp(layer.w), vp(layer.bias), vp(out), vp(res), vp(add)
    p.x_bstride, p.out_bstride, p.res_bstride, p.add_bstride = x_bs, out_bs, res_bs, add_bs
    p.B, p.Cin, p.Cout, p.L, p.x_ld, p.out_ld = B, layer.cin, layer.cout, L, x_ld, out_ld
    p.K, p.dil, p.nwg, p.frags = layer.K, layer.dil, nwg, frags
    p.in_slope, p.scale, p.out_slope = in_slope, scale, out_slope
    p.dbg = vp(dbg)
    p.col_limit, p.col_limit_scale = vp(col_limit), col_limit_scale
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_wino_f32(ctypes.byref(p), stream), "ov_conv1d_wino_f32")
