"""Host helpers of the bf16 channels-last generator kernels (csrc/conv1d_bf16.hip): weight packing and launch.
Activations are ``torch.bfloat16`` tensors of shape (B, L, C) -- channels contiguous."""
import ctypes

import torch

from . import _lib
from ._lib import ConvBf16Params


class PackedConvBf16:
    """One conv layer in bf16 kernel-ready form: B-fragment-ordered bf16 weights + fp32 bias."""

    def __init__(self, w_dense, bias, device, dil=1):
        lib = _lib.load()
        w = w_dense.detach().to(torch.float32).cpu().contiguous()
        self.cout, self.cin, self.K = w.shape
        self.dil = dil
        n = lib.ov_conv1d_bf16_pack_size(self.cout, self.cin, self.K)
        if n == 0:
            raise _lib.OvError(f"bf16 conv needs Cin % 32 == 0 (got {self.cin})")
        packed = torch.empty(n, dtype=torch.int16)
        _lib.check(lib.ov_conv1d_bf16_pack(ctypes.c_void_p(w.data_ptr()), self.cout, self.cin, self.K,
                                           ctypes.c_void_p(packed.data_ptr())), "ov_conv1d_bf16_pack")
        self.w = packed.to(device)
        self.bias = None if bias is None else bias.detach().float().contiguous().to(device)


def launch_conv_bf16(layer, x, out, in_slope=1.0, scale=1.0, res=None, add=None):
    """out = (conv1d(lrelu(x, in_slope)) + bias [+ res] [+ add]) * scale on torch's current stream.
    x (B, L, Cin), out / res / add (B, L, Cout), all contiguous bfloat16."""
    B, L, cin = x.shape
    assert cin == layer.cin and out.shape == (B, L, layer.cout)
    for t in (x, out, res, add):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous())
    p = ConvBf16Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w, p.bias, p.out, p.res, p.add = vp(x), vp(layer.w), vp(layer.bias), vp(out), vp(res), vp(add)
    p.B, p.L, p.Cin, p.Cout, p.K, p.dil = B, L, cin, layer.cout, layer.K, layer.dil
    p.in_slope, p.scale = in_slope, scale
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_bf16cl(ctypes.byref(p), stream), "ov_conv1d_bf16cl")
