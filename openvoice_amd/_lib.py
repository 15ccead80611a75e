"""ctypes binding of libopenvoice_amd.so (the C ABI declared in include/openvoice_amd.h).

There is no CPU fallback: if the shared library is missing or a launch fails, the caller gets
an exception.  ``python -c "import __graft_entry__ as g; g.build()"`` (or
``make -C openvoice_amd/csrc``) produces the library in-tree.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPENVOICE_AMD_LIB: measurement builds (scripts/exp_sync.sh) live outside the tree and are selected explicitly
LIB_PATH = os.environ.get("OPENVOICE_AMD_LIB") or os.path.join(_HERE, "libopenvoice_amd.so")

OV_OK = 0
OV_ERRORS = {-1: "OV_E_BADARG", -2: "OV_E_UNSUPPORTED", -3: "OV_E_ALIGN", -4: "OV_E_LAUNCH"}

EPI_LINEAR, EPI_GATE, EPI_RESSKIP, EPI_COUPLE, EPI_POSTERIOR, EPI_CONVT, EPI_MAGNITUDE = range(7)
F_MASK_V = 1
F_OUT2_INIT = 2
F_CONVT_GROUPED = 4
F_NO_XCD_MAP = 8      # measurement knob: round-robin tile order instead of XCD-contiguous (include/openvoice_amd.h)

_fp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
LN_PRE_RELU, LN_POST_GELU = 1, 2


class ConvParams(ctypes.Structure):
    """Mirror of ``ov_conv1d_params`` (include/openvoice_amd.h)."""
    _fields_ = [
        ("x", _fp), ("w", _fp), ("bias", _fp), ("bias_b", _fp), ("out", _fp), ("res", _fp),
        ("add", _fp), ("out2", _fp), ("mask", _fp),
        ("x_bstride", ctypes.c_int64), ("out_bstride", ctypes.c_int64), ("res_bstride", ctypes.c_int64),
        ("add_bstride", ctypes.c_int64), ("out2_bstride", ctypes.c_int64), ("bias_b_bstride", ctypes.c_int64),
        ("mask_bstride", ctypes.c_int64),
        ("B", ctypes.c_int32), ("Cin", ctypes.c_int32), ("L", ctypes.c_int32),
        ("x_ld", ctypes.c_int32), ("out_ld", ctypes.c_int32), ("M", ctypes.c_int32),
        ("Cout", ctypes.c_int32), ("K", ctypes.c_int32), ("dil", ctypes.c_int32), ("epi", ctypes.c_int32),
        ("flags", ctypes.c_int32), ("split", ctypes.c_int32), ("phase_s", ctypes.c_int32),
        ("tiles_per_wg", ctypes.c_int32), ("tile", ctypes.c_int32), ("loaders", ctypes.c_int32),
        ("chunk", ctypes.c_int32),
        ("in_slope", ctypes.c_float), ("scale", ctypes.c_float),
    ]


class ConvBf16Params(ctypes.Structure):
    """Mirror of ``ov_conv1d_bf16_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w", _fp), ("bias", _fp), ("out", _fp), ("res", _fp), ("add", _fp),
                ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("K", ctypes.c_int32), ("dil", ctypes.c_int32), ("phase_s", ctypes.c_int32), ("bias_bstride", ctypes.c_int32), ("layout", ctypes.c_int32),
                ("in_slope", ctypes.c_float), ("scale", ctypes.c_float), ("out_slope", ctypes.c_float), ("dbg", _fp)]


class RespairParams(ctypes.Structure):
    """Mirror of ``ov_respair_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("out", _fp), ("add", _fp),
                ("x_bstride", ctypes.c_int64), ("out_bstride", ctypes.c_int64), ("add_bstride", ctypes.c_int64),
                ("B", ctypes.c_int32), ("C", ctypes.c_int32), ("L", ctypes.c_int32), ("ld", ctypes.c_int32),
                ("K", ctypes.c_int32), ("dil", ctypes.c_int32), ("nwg", ctypes.c_int32),
                ("slope", ctypes.c_float), ("scale", ctypes.c_float), ("dbg", _fp)]


class RespairBf16Params(ctypes.Structure):
    """Mirror of ``ov_respair_bf16_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("out", _fp), ("add", _fp),
                ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32),
                ("dil", ctypes.c_int32), ("nwg", ctypes.c_int32), ("slope", ctypes.c_float), ("scale", ctypes.c_float),
                ("dbg", _fp)]


class WnLayerParams(ctypes.Structure):
    """Mirror of ``ov_wn_layer_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("out", _fp), ("skip", _fp), ("w_in", _fp), ("b_in", _fp), ("cond", _fp), ("w_rs", _fp),
                ("b_rs", _fp), ("mask", _fp),
                ("bstride", ctypes.c_int64), ("cond_bstride", ctypes.c_int64), ("mask_bstride", ctypes.c_int64),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("T", ctypes.c_int32), ("ld", ctypes.c_int32),
                ("K", ctypes.c_int32), ("first", ctypes.c_int32), ("last", ctypes.c_int32), ("width", ctypes.c_int32),
                ("ntile", ctypes.c_int32), ("reserved", ctypes.c_int32), ("dbg", _fp)]


# name -> (restype, argtypes); kept in one table so tests can check every header symbol exists.
SIGNATURES = {
    "ov_version": (ctypes.c_int, []),
    "ov_build_experiment": (ctypes.c_int, []),
    "ov_conv1d_pack_size": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ov_conv1d_pack_rows": (ctypes.c_int, [ctypes.c_int]),
    "ov_conv1d_pack_f32": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_conv1d_f32": (ctypes.c_int, [ctypes.POINTER(ConvParams), _fp]),
    "ov_resblock_pair_f32": (ctypes.c_int, [ctypes.POINTER(RespairParams), _fp]),
    "ov_resblock_pair_supported": (ctypes.c_int, [_i, _i, _i]),
    "ov_wn_layer_f32": (ctypes.c_int, [ctypes.POINTER(WnLayerParams), _fp]),
    "ov_wn_layer_supported": (ctypes.c_int, [_i, _i]),
    "ov_wn_pack_size": (ctypes.c_size_t, [_i, _i, _i]),
    "ov_wn_pack_f32": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
    "ov_wn_layer_tile": (ctypes.c_int, [_i, _i, _i]),
    "ov_conv_post_tanh_f32": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_float, _fp]),
    "ov_linear_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_sequence_mask_f32": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_layernorm_freq_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, _fp]),
    "ov_conv2d_s2_relu_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, _fp]),
    "ov_gru_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_conv1d_bf16_pack_size": (ctypes.c_size_t, [_i, _i, _i]),
    "ov_conv1d_bf16_pack": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
    "ov_conv1d_bf16cl": (ctypes.c_int, [ctypes.POINTER(ConvBf16Params), _fp]),
    "ov_resblock_pair_bf16cl": (ctypes.c_int, [ctypes.POINTER(RespairBf16Params), _fp]),
    "ov_resblock_pair_bf16_supported": (ctypes.c_int, [_i, _i, _i]),
    "ov_conv_post_tanh_bf16": (ctypes.c_int, [_fp, _fp, _fp, _i, _i, _i, _i, ctypes.c_float, _fp]),
    "ov_frame_hops_f32": (ctypes.c_int, [_fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    "ov_embed_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, ctypes.c_float, _fp]),
    "ov_layernorm_ch_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, ctypes.c_float, _i, _fp]),
    "ov_rel_attention_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _i64, _i, _i, _i, _i, _i, _i,
                                            _fp]),
    "ov_add_bias_mask_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    "ov_dwconv1d_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    "ov_expand1_f32": (ctypes.c_int, [_fp, _i64, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    "ov_rq_spline_inverse_f32": (ctypes.c_int, [_fp, _i64, _i, _i, _fp, _i64, _fp, _i, _i, _i, _i, _i,
                                                ctypes.c_float, _fp]),
    "ov_duration_f32": (ctypes.c_int, [_fp, _i64, ctypes.c_float, ctypes.c_float, _fp, _i64, _fp, _fp, _fp, _fp,
                                       _i, _i, _i, ctypes.c_float, ctypes.c_float, _fp]),
    "ov_expand_prior_f32": (ctypes.c_int, [_fp, _fp, _i64, _i, _fp, _fp, _fp, _fp, _i64, _i, _fp, _fp, _fp, _fp,
                                           _i, _i, _i, _i, _i, ctypes.c_float, _fp]),
}

_lib = None


class OvError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OvError(f"{LIB_PATH} not found: build the HIP extension first "
                          f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        exp = lib.ov_build_experiment()
        if exp != 0 and os.environ.get("OPENVOICE_AMD_ALLOW_EXPERIMENT") != "1":
            raise OvError(f"{LIB_PATH} is a measurement build (OV_EXP={exp}: kernels with loads / barriers compiled "
                          f"out, results meaningless); rebuild with `make -C openvoice_amd/csrc clean all`")
        _lib = lib
    return _lib


def check(code, what):
    if code != OV_OK:
        raise OvError(f"{what} failed: {OV_ERRORS.get(code, code)}")


# ---- the torch binding (csrc/torch_shim.cpp): `torch.ops.openvoice_amd.*` -------------------------------------------
SHIM_PATH = os.path.join(_HERE, "libopenvoice_amd_torch.so")
_ops = None


def use_torch_binding():
    """True when launches go through ``torch.ops.openvoice_amd`` (TORCH_LIBRARY shim: current-stream pickup,
    TORCH_CHECK errors) instead of ctypes.  Selected with OPENVOICE_AMD_BINDING=torch; both bindings drive the same
    C ABI with the same arguments."""
    return os.environ.get("OPENVOICE_AMD_BINDING", "ctypes") == "torch"


def torch_ops():
    """Load (once) the shim and return ``torch.ops.openvoice_amd``; raises if it was not built."""
    global _ops
    if _ops is None:
        import torch
        load()                       # libopenvoice_amd.so first: the shim links against it by soname
        if not os.path.exists(SHIM_PATH):
            raise OvError(f"{SHIM_PATH} not found: make -C openvoice_amd/csrc torch_shim")
        torch.ops.load_library(SHIM_PATH)
        _ops = torch.ops.openvoice_amd
    return _ops
