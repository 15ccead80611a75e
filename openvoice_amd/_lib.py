"""The two bindings of libopenvoice_amd.so (the C ABI declared in include/openvoice_amd.h).

* ``torch`` (default; BASELINE.json north_star "bound through a thin torch cpp_extension C-ABI"): every launch goes
  through ``torch.ops.openvoice_amd.*`` (csrc/torch_shim.cpp -> libopenvoice_amd_torch.so, a TORCH_LIBRARY shim with one
  op per C entry point: tensors in, device / dtype TORCH_CHECKs, current HIP stream picked up in C++, OV_E_* codes
  raised as RuntimeError).  ctypes is not touched at all in this mode (tests/test_gpu_torch_shim.py makes ``load()``
  raise and runs a conversion, ``extract_se`` and ``infer``).
* ``ctypes`` (``OPENVOICE_AMD_BINDING=ctypes``): the same entry points through ``ctypes.CDLL`` -- what a consumer
  without libtorch binds (INTEGRATION.md level C), and what the C-ABI tests use to reach the functions directly.

Both drive the same kernels with the same arguments; ``call()`` is the one switch point for every entry point with a
flat argument list, the five parameter-struct entry points have a launch helper each (engine.py, bf16.py).

There is no CPU fallback: if the shared libraries are missing or a launch fails, the caller gets an exception.
``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C openvoice_amd/csrc``) produces them in-tree.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPENVOICE_AMD_LIB: measurement builds (scripts/exp_sync.sh) live outside the tree and are selected explicitly
LIB_PATH = os.environ.get("OPENVOICE_AMD_LIB") or os.path.join(_HERE, "libopenvoice_amd.so")

OV_OK = 0
MIN_VERSION = 209     # ov_version() of the newest entry point this package calls (include/openvoice_amd.h)
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
        ("col_limit", _fp), ("col_limit_scale", ctypes.c_int32), ("reserved0", ctypes.c_int32),
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
                ("slope", ctypes.c_float), ("scale", ctypes.c_float), ("dbg", _fp),
                ("col_limit", _fp), ("col_limit_scale", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class RespairBf16Params(ctypes.Structure):
    """Mirror of ``ov_respair_bf16_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("out", _fp), ("add", _fp),
                ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32),
                ("dil", ctypes.c_int32), ("nwg", ctypes.c_int32), ("slope", ctypes.c_float), ("scale", ctypes.c_float),
                ("dbg", _fp)]


class Respair2Bf16Params(ctypes.Structure):
    """Mirror of ``ov_respair2_bf16_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("out", _fp), ("add", _fp),
                ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32),
                ("dil", ctypes.c_int32), ("nwg", ctypes.c_int32), ("slope", ctypes.c_float), ("scale", ctypes.c_float),
                ("out_slope", ctypes.c_float), ("exp_flags", ctypes.c_int32), ("dbg", _fp)]


class ConvSplit3Params(ctypes.Structure):
    """Mirror of ``ov_conv1d_split3_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w", _fp), ("bias", _fp), ("out", _fp), ("res", _fp),
                ("x_plane", ctypes.c_int64), ("out_plane", ctypes.c_int64), ("res_plane", ctypes.c_int64),
                ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("K", ctypes.c_int32), ("dil", ctypes.c_int32), ("nwg", ctypes.c_int32), ("products", ctypes.c_int32),
                ("res_slope", ctypes.c_float), ("out_slope", ctypes.c_float), ("scale", ctypes.c_float),
                ("col_limit_scale", ctypes.c_int32), ("dbg", _fp), ("col_limit", _fp)]


class ConvWinoParams(ctypes.Structure):
    """Mirror of ``ov_conv1d_wino_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("w", _fp), ("bias", _fp), ("out", _fp), ("res", _fp), ("add", _fp),
                ("x_bstride", ctypes.c_int64), ("out_bstride", ctypes.c_int64), ("res_bstride", ctypes.c_int64),
                ("add_bstride", ctypes.c_int64),
                ("B", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32), ("L", ctypes.c_int32),
                ("x_ld", ctypes.c_int32), ("out_ld", ctypes.c_int32), ("K", ctypes.c_int32), ("dil", ctypes.c_int32),
                ("nwg", ctypes.c_int32), ("frags", ctypes.c_int32),
                ("in_slope", ctypes.c_float), ("scale", ctypes.c_float), ("dbg", _fp),
                ("col_limit", _fp), ("col_limit_scale", ctypes.c_int32), ("out_slope", ctypes.c_float)]


class WnLayerParams(ctypes.Structure):
    """Mirror of ``ov_wn_layer_params`` (include/openvoice_amd.h)."""
    _fields_ = [("x", _fp), ("out", _fp), ("skip", _fp), ("w_in", _fp), ("b_in", _fp), ("cond", _fp), ("w_rs", _fp),
                ("b_rs", _fp), ("mask", _fp),
                ("bstride", ctypes.c_int64), ("cond_bstride", ctypes.c_int64), ("mask_bstride", ctypes.c_int64),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("T", ctypes.c_int32), ("ld", ctypes.c_int32),
                ("K", ctypes.c_int32), ("first", ctypes.c_int32), ("last", ctypes.c_int32), ("width", ctypes.c_int32),
                ("ntile", ctypes.c_int32), ("row_split", ctypes.c_int32), ("dbg", _fp), ("acts", _fp)]


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
    "ov_conv_post_tanh_limited_f32": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_float, _fp, ctypes.c_int, _fp]),
    "ov_frame_limits_i32": (ctypes.c_int, [_fp, _fp, _i, _i, _i, _fp]),
    "ov_linear_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_sequence_mask_f32": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_unpad_rows_f32": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_polyphase_fir_f32": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int64, ctypes.c_int64, _i, _i, _i, _fp]),
    "ov_layernorm_freq_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, _fp]),
    "ov_conv2d_s2_relu_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, _fp]),
    "ov_gru_f32": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    "ov_conv1d_bf16_pack_size": (ctypes.c_size_t, [_i, _i, _i]),
    "ov_conv1d_bf16_pack": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
    "ov_conv1d_bf16_pack16": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
    "ov_conv1d_bf16cl": (ctypes.c_int, [ctypes.POINTER(ConvBf16Params), _fp]),
    "ov_resblock_pair_bf16cl": (ctypes.c_int, [ctypes.POINTER(RespairBf16Params), _fp]),
    "ov_resblock_pair_bf16_supported": (ctypes.c_int, [_i, _i, _i]),
    "ov_resblock_pair2_bf16cl": (ctypes.c_int, [ctypes.POINTER(Respair2Bf16Params), _fp]),
    "ov_resblock_pair2_bf16_supported": (ctypes.c_int, [_i, _i, _i]),
    "ov_conv_post_tanh_bf16": (ctypes.c_int, [_fp, _fp, _fp, _i, _i, _i, _i, ctypes.c_float, _fp]),
    "ov_conv1d_split3": (ctypes.c_int, [ctypes.POINTER(ConvSplit3Params), _fp]),
    "ov_conv1d_split3_pack_size": (ctypes.c_size_t, [_i, _i, _i]),
    "ov_conv1d_split3_pack": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
    "ov_conv1d_split3_supported": (ctypes.c_int, [_i, _i, _i, _i]),
    "ov_split3_from_f32": (ctypes.c_int, [_fp, _fp, _i64, _i, _i, _i, ctypes.c_float, _fp]),
    "ov_split3_to_f32": (ctypes.c_int, [_fp, _fp, _fp, _i64, _fp, _i, _i, _i, ctypes.c_float, ctypes.c_float, _fp]),
    "ov_conv1d_wino_f32": (ctypes.c_int, [ctypes.POINTER(ConvWinoParams), _fp]),
    "ov_conv1d_wino_supported": (ctypes.c_int, [_i, _i, _i, _i]),
    "ov_conv1d_wino_chunk": (ctypes.c_int, [_i, _i]),
    "ov_conv1d_wino_pack_size": (ctypes.c_size_t, [_i, _i, _i]),
    "ov_conv1d_wino_pack_f32": (ctypes.c_int, [_fp, _i, _i, _i, _fp]),
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
        lib.ov_version.restype, lib.ov_version.argtypes = ctypes.c_int, []
        if lib.ov_version() < MIN_VERSION:       # before the symbol lookups: a stale library fails HERE, by version
            raise OvError(f"{LIB_PATH} is version {lib.ov_version()}, this package needs >= {MIN_VERSION}: rebuild it "
                          f"(make -C openvoice_amd/csrc)")
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


def torch_op(name, *args):
    """One ``torch.ops.openvoice_amd`` op (the parameter-struct launch helpers); errors as ``OvError``."""
    try:
        return getattr(torch_ops(), name)(*args)
    except RuntimeError as exc:
        if isinstance(exc, OvError):
            raise
        raise OvError(str(exc).split("\n")[0]) from None


def check(code, what):
    if code != OV_OK:
        raise OvError(f"{what} failed: {OV_ERRORS.get(code, code)}")


# ---- the torch binding (csrc/torch_shim.cpp): `torch.ops.openvoice_amd.*` -------------------------------------------
SHIM_PATH = os.path.join(_HERE, "libopenvoice_amd_torch.so")
_ops = None
# entry points whose return value is a quantity, not an OV_* status
VALUE_FUNCS = {"ov_version", "ov_build_experiment", "ov_conv1d_pack_size", "ov_conv1d_pack_rows", "ov_wn_pack_size",
               "ov_conv1d_bf16_pack_size", "ov_resblock_pair_supported", "ov_wn_layer_supported", "ov_wn_layer_tile",
               "ov_resblock_pair_bf16_supported", "ov_resblock_pair2_bf16_supported", "ov_conv1d_split3_pack_size",
               "ov_conv1d_split3_supported", "ov_conv1d_wino_supported", "ov_conv1d_wino_chunk", "ov_conv1d_wino_pack_size"}


def binding():
    """"torch" (default) or "ctypes" -- read from OPENVOICE_AMD_BINDING at every call, so a test can switch it."""
    # a measurement build selected with OPENVOICE_AMD_LIB lives outside the tree; only ctypes can load it
    b = os.environ.get("OPENVOICE_AMD_BINDING", "ctypes" if os.environ.get("OPENVOICE_AMD_LIB") else "torch")
    if b not in ("torch", "ctypes"):
        raise OvError(f"OPENVOICE_AMD_BINDING={b!r}: expected 'torch' or 'ctypes'")
    return b


def use_torch_binding():
    """True when launches go through ``torch.ops.openvoice_amd`` (the TORCH_LIBRARY shim) instead of ctypes."""
    return binding() == "torch"


def torch_ops():
    """Load (once) the shim and return ``torch.ops.openvoice_amd``; raises if it was not built.  The shim links
    libopenvoice_amd.so through its ``$ORIGIN`` rpath: ctypes is not involved."""
    global _ops
    if _ops is None:
        import torch
        for path in (LIB_PATH, SHIM_PATH):
            if not os.path.exists(path):
                raise OvError(f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ "
                              f"as g; g.build()'); there is no CPU fallback (OPENVOICE_AMD_BINDING=ctypes needs only "
                              f"libopenvoice_amd.so)")
        torch.ops.load_library(SHIM_PATH)
        ops = torch.ops.openvoice_amd
        if ops.version() < MIN_VERSION:
            raise OvError(f"{LIB_PATH} is version {ops.version()}, this package needs >= {MIN_VERSION}: rebuild it "
                          f"(make -C openvoice_amd/csrc)")
        exp = ops.build_experiment()
        if exp != 0 and os.environ.get("OPENVOICE_AMD_ALLOW_EXPERIMENT") != "1":
            raise OvError(f"{LIB_PATH} is a measurement build (OV_EXP={exp}); rebuild with "
                          f"`make -C openvoice_amd/csrc clean all`")
        _ops = ops
    return _ops


def _flat_view(t, off):
    """A 1-D alias of ``t``'s storage starting ``off`` elements after its first element (what ``ptr + off`` is to C)."""
    start = t.storage_offset() + off
    n = t.untyped_storage().nbytes() // t.element_size() - start
    return t.as_strided((n,), (1,), start)


def call(name, *args):
    """Invoke the C entry point ``name`` (e.g. ``"ov_linear_f32"``) through the selected binding.  ``args`` follow
    the C prototype WITHOUT the trailing stream: tensors (or ``(tensor, element_offset)`` pairs, or None) where C takes
    pointers, Python numbers elsewhere.  Device entry points launch on torch's current stream of the tensors' device.
    Status-returning functions raise on a non-zero code; size / capability queries return their value."""
    if use_torch_binding():
        targs = [_flat_view(*a) if type(a) is tuple else a for a in args]
        try:
            return getattr(torch_ops(), name[3:])(*targs)
        except RuntimeError as exc:       # TORCH_CHECK failures surface as the same exception type as ctypes codes
            if isinstance(exc, OvError):
                raise
            raise OvError(str(exc).split("\n")[0]) from None
    import torch
    lib = load()
    fn = getattr(lib, name)
    cargs, dev = [], None
    for a in args:
        off = 0
        if type(a) is tuple:
            a, off = a
        if isinstance(a, torch.Tensor):
            if dev is None and a.is_cuda:
                dev = a.device
            a = ctypes.c_void_p(a.data_ptr() + off * a.element_size())
        cargs.append(a)
    if len(fn.argtypes) == len(cargs) + 1:            # device entry point: the stream is the last C parameter
        if dev is None:
            raise OvError(f"{name}: no device tensor among the arguments")
        cargs.append(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    rc = fn(*cargs)
    if name in VALUE_FUNCS:
        return rc
    check(rc, name)
