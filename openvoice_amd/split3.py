"""Host helpers of the split-precision conv kernels (csrc/conv1d_split3.h): fp32-level products on the bf16 matrix pipe.

An fp32 tensor is carried as three bf16 planes ``v = hi + mid + lo`` (lossless), channels-last and plane-major:
``torch.bfloat16`` tensors of shape (3, B, L, C).  ``PackedConvSplit3`` holds a conv's weights as three planes in MFMA
fragment order; ``launch_conv_split3`` runs one conv with its residual / scale / activation epilogue; ``to_planes`` /
``from_planes`` are the layout kernels between the fp32 engine's (B, C, L) tensors and the planes.
Opt-in (``ConverterEngine.use_split_bf16x3()``); the default fp32 path never comes here.
reference: openvoice/modules.py:296-309 (ResBlock1.forward), models.py:280-286 (MRF mean)."""
import ctypes

import torch

from . import _lib


class PackedConvSplit3:
    """One conv layer in split-precision kernel-ready form: three bf16 weight planes in 16x16x32 A-fragment order
    (``ov_conv1d_split3_pack``) + fp32 bias."""

    def __init__(self, w_dense, bias, device, dil=1):
        w = w_dense.detach().to(torch.float32).cpu().contiguous()
        self.cout, self.cin, self.K = w.shape
        self.dil = dil
        if not _lib.call("ov_conv1d_split3_supported", self.cin, self.cout, self.K, dil):
            raise _lib.OvError(f"ov_conv1d_split3: no instance for Cin={self.cin} Cout={self.cout} K={self.K} dil={dil}")
        n = _lib.call("ov_conv1d_split3_pack_size", self.cout, self.cin, self.K)
        packed = torch.empty(n, dtype=torch.int16)
        _lib.call("ov_conv1d_split3_pack", w, self.cout, self.cin, self.K, packed)
        self.w = packed.to(device)
        b = torch.zeros(self.cout) if bias is None else bias.detach().float()
        self.bias = b.contiguous().to(device)


def supported(cin, cout, K, dil):
    return bool(_lib.call("ov_conv1d_split3_supported", cin, cout, K, dil))


def launch_conv_split3(layer, x, out, res=None, res_slope=1.0, out_slope=1.0, scale=1.0, products=6, nwg=0, dbg=None,
                       col_limit=None, col_limit_scale=1):
    """out = split3(lrelu((conv1d(x) + bias [+ res~]) * scale, out_slope)) on torch's current stream.  x (3, B, L, Cin),
    out / res (3, B, L, Cout): contiguous bfloat16 plane tensors; ``x`` is read as stored (activated by its producer),
    res~ = the inverse leaky ReLU of ``res`` with ``res_slope``.  ``col_limit`` (int32 [B] on the device) x
    ``col_limit_scale`` = columns of each utterance that matter: time tiles beyond are neither computed nor written."""
    _, B, L, cin = x.shape
    assert x.shape[0] == 3 and cin == layer.cin and out.shape == (3, B, L, layer.cout)
    for t in (x, out, res):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous() and t.shape[0] == 3)
    xp, op = B * L * cin, B * L * layer.cout
    if _lib.use_torch_binding():
        _lib.torch_op("conv1d_split3", x, layer.w, layer.bias, out, res, dbg,
                      col_limit, [B, L, cin, layer.cout, layer.K, layer.dil, nwg, products, xp, op,
                                  op if res is not None else 0, col_limit_scale], [res_slope, out_slope, scale])
        return
    p = _lib.ConvSplit3Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w, p.bias, p.out, p.res = vp(x), vp(layer.w), vp(layer.bias), vp(out), vp(res)
    p.x_plane, p.out_plane, p.res_plane = xp, op, op if res is not None else 0
    p.B, p.L, p.Cin, p.Cout, p.K, p.dil, p.nwg, p.products = B, L, cin, layer.cout, layer.K, layer.dil, nwg, products
    p.res_slope, p.out_slope, p.scale = res_slope, out_slope, scale
    p.dbg, p.col_limit, p.col_limit_scale = vp(dbg), vp(col_limit), col_limit_scale
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_split3(ctypes.byref(p), stream), "ov_conv1d_split3")


def to_planes(x, slope=1.0, out=None):
    """(B, C, L) fp32 (dense) -> planes (3, B, L, C) of lrelu(x, slope) (``ov_split3_from_f32``)."""
    B, C, L = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(3, B, L, C, dtype=torch.bfloat16, device=x.device)
    _lib.call("ov_split3_from_f32", x, out, B * L * C, B, C, L, slope)
    return out


def from_planes(a, b=None, c=None, in_slope=1.0, scale=1.0, out=None):
    """(a~ [+ b~] [+ c~]) * scale as (B, C, L) fp32; each operand a (3, B, L, C) plane tensor stored activated with
    ``in_slope`` (``ov_split3_to_f32``)."""
    _, B, L, C = a.shape
    if out is None:
        out = torch.empty(B, C, L, dtype=torch.float32, device=a.device)
    _lib.call("ov_split3_to_f32", a, b, c, B * L * C, out, B, C, L, in_slope, scale)
    return out


def split3_reference(v):
    """The plane split in plain torch (any device): (3, ...) bfloat16 with hi + mid + lo == v exactly."""
    hi = v.to(torch.bfloat16)
    r = v - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo])
