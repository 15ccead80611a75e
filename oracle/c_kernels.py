"""ORACLE (test infrastructure): ctypes access to the plain-C restatement ``vc_kernels_ref.c`` -- loop-level conv1d /
conv-transpose / WaveNet gate / WaveNet layer exactly as the reference evaluates them (file:line in the C source).
Built by ``oracle/Makefile`` (``__graft_entry__.build()`` runs it; built on demand here if missing)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvc_kernels_ref.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "vc_kernels_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = ctypes.CDLL(_SO)
        for name in ("ref_conv1d_f32", "ref_conv_transpose1d_f32", "ref_gate_f32", "ref_wn_layer_f32",
                     "ref_rq_spline_inverse_f32", "ref_rel_attention_f32"):
            getattr(_lib, name).restype = None
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def conv1d(x, w, bias, dil=1, slope=1.0):
    """``conv1d(leaky_relu(x, slope), w, bias, dilation=dil, padding='same')`` (modules.py:296-306, commons.py:12-13)."""
    x, w, bias = _f(x), _f(w), _f(bias)
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    y = torch.empty(B, Cout, L)
    load().ref_conv1d_f32(_p(x), _p(w), _p(bias), _p(y), B, Cin, Cout, L, K, dil, ctypes.c_float(slope))
    return y


def conv_transpose1d(x, w, bias, stride, pad, slope=1.0):
    """``conv_transpose1d(leaky_relu(x, slope), w, bias, stride, padding=pad)`` (models.py:244-256, 278-279)."""
    x, w, bias = _f(x), _f(w), _f(bias)
    B, Cin, L = x.shape
    _, Cout, K = w.shape
    y = torch.empty(B, Cout, (L - 1) * stride - 2 * pad + K)
    load().ref_conv_transpose1d_f32(_p(x), _p(w), _p(bias), _p(y), B, Cin, Cout, L, K, stride, pad, ctypes.c_float(slope))
    return y


def gate(x_in, g):
    """``fused_add_tanh_sigmoid_multiply(x_in, g_l, [H])`` with ``g`` [B, 2H] (commons.py:100-107)."""
    x_in, g = _f(x_in), _f(g)
    B, H2, T = x_in.shape
    acts = torch.empty(B, H2 // 2, T)
    load().ref_gate_f32(_p(x_in), _p(g), _p(acts), B, H2 // 2, T)
    return acts


def wn_layer(x, output, w_in, b_in, w_rs, b_rs, g, mask, dil, last):
    """One layer of ``WN.forward`` (modules.py:192-209) in place on ``x`` / ``output`` (both [B, H, T], contiguous)."""
    B, H, T = x.shape
    assert x.is_contiguous() and output.is_contiguous() and x.dtype == output.dtype == torch.float32
    w_in, b_in, w_rs, b_rs, g, mask = (_f(t) for t in (w_in, b_in, w_rs, b_rs, g, mask))
    scratch = torch.empty(5 * H * T)
    load().ref_wn_layer_f32(_p(x), _p(output), _p(w_in), _p(b_in), _p(w_rs), _p(b_rs), _p(g), _p(mask), B, H, T,
                            w_in.shape[2], dil, 1 if last else 0, _p(scratch))


def rq_spline_inverse(y, uw, uh, ud, tail_bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """Element-wise inverse rational-quadratic spline with linear tails (transforms.py:50-188); ``y`` [...],
    ``uw`` / ``uh`` [..., bins], ``ud`` [..., bins - 1]."""
    y, uw, uh, ud = _f(y), _f(uw), _f(uh), _f(ud)
    nb = uw.shape[-1]
    x = torch.empty_like(y)
    load().ref_rq_spline_inverse_f32(_p(y), _p(uw), _p(uh), _p(ud), _p(x), ctypes.c_long(y.numel()), nb,
                                     ctypes.c_float(tail_bound), ctypes.c_float(min_w), ctypes.c_float(min_h),
                                     ctypes.c_float(min_d))
    return x


def rel_attention(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window):
    """Attention core of ``MultiHeadAttention`` with relative keys / values (attentions.py:264-329); ``q``/``k``/``v``
    [B, C, T] straight out of conv_q / conv_k / conv_v, ``emb_rel_*`` [2 * window + 1, C // n_heads], ``mask`` [B, T]."""
    q, k, v, ek, ev, mask = (_f(t) for t in (q, k, v, emb_rel_k, emb_rel_v, mask))
    B, C, T = q.shape
    out, row = torch.empty_like(q), torch.empty(T)
    load().ref_rel_attention_f32(_p(q), _p(k), _p(v), _p(ek), _p(ev), _p(mask), _p(out), B, n_heads, C // n_heads, T,
                                 window, _p(row))
    return out
