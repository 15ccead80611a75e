"""Host-side driver of the MI355X tone-colour-converter kernels.

``ConverterEngine`` owns the load-time weight transforms (weight-norm folding, Flip folding,
gate/posterior row pairing, transposed-conv phase split, MFMA fragment packing) and issues the
kernel launches of one ``voice_conversion`` call through the C ABI (include/openvoice_amd.h) on
torch's current HIP stream.  PyTorch is used for device memory and streams only: every op between
``spec [B,513,T]`` and ``o_hat [B,1,256T]`` is a kernel from libopenvoice_amd.so.

Reference path being replaced: SynthesizerTrn.voice_conversion (openvoice/models.py:492-499) =
PosteriorEncoder (models.py:212-221) -> ResidualCouplingBlock forward with g_src and reverse with
g_tgt (models.py:390-397) -> Generator (models.py:272-291).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ConvParams, EPI_CONVT, EPI_COUPLE, EPI_GATE, EPI_LINEAR, EPI_POSTERIOR, EPI_RESSKIP,
                   F_CONVT_GROUPED, F_MASK_V, F_OUT2_INIT)
from .params import ENC_Q_LAYERS, FLOW_LAYERS, N_FLOWS, REF_ENC_FILTERS, REF_ENC_GRU, effective_weight

# Length-aware work lists (``skip_padding``): the generator's one-sided receptive field is 13.3 frames -- conv_pre 3,
# the four transposed convs 1 + 1/8 + 1/64 + 1/128, the MRF 60 samples per stage (k = 11: dilations 1, 3, 5 + three
# dilation-1 convs = 5 * (1 + 3 + 5 + 3)) at 8 / 64 / 128 / 256 samples per frame, conv_post 3 samples (reference:
# openvoice/models.py:225-291, modules.py:221-309) -- so computing length + 16 frames leaves the first ``length``
# frames bit-identical to the full computation.
GENERATOR_MARGIN = 16   # the released configuration's; every engine computes its own (generator_margin_frames)
LIMIT_MAX_BATCH = 256   # ovk::LIMIT_MAX_BATCH: utterances per launch the kernels' prefix table holds
LRELU_SLOPE = 0.1       # reference: openvoice/modules.py:14
FINAL_LRELU_SLOPE = 0.01  # F.leaky_relu default at openvoice/models.py:287


# What the kernel library has instances for (csrc/conv1d_inst_*.hip; everything else returns OV_E_UNSUPPORTED at the
# first launch -- validate_config says so at load time, by field name, before any weight is packed).
SUPPORTED_RESBLOCK_KERNELS = (3, 7, 11)
SUPPORTED_RESBLOCK_DILATIONS = (1, 3, 5)


def validate_config(cfg):
    """Reject, naming the field, every hyper-parameter the kernels have no instance for.  The reference is
    config-driven (openvoice/models.py:225-270: ``resblock`` picks ResBlock1 / ResBlock2 at :242, kernel sizes, dilations
    and upsampling rates come from the JSON); this library instantiates the released family: ResBlock1, MRF kernels
    3 / 7 / 11 with dilations 1 / 3 / 5, ConvTranspose1d with kernel = 2 x stride (stride a divisor of 32), channel
    counts in whole 32-row MFMA fragments."""
    def bad(field, value, why):
        raise _lib.OvError(f"config field {field!r} = {value!r} is not supported by the MI355X kernels: {why}")

    rb = str(cfg.get("resblock", "1"))
    if rb != "1":
        bad("resblock", cfg.get("resblock"), "only ResBlock1 ('1') is built (ResBlock2 is not; reference "
            "openvoice/models.py:242, modules.py:312-357)")
    kernels, dils = list(cfg["resblock_kernel_sizes"]), [list(d) for d in cfg["resblock_dilation_sizes"]]
    if len(kernels) != len(dils) or not kernels:
        bad("resblock_dilation_sizes", cfg["resblock_dilation_sizes"], f"one dilation list per entry of "
            f"resblock_kernel_sizes ({len(kernels)}) is required")
    for k in kernels:
        if k not in SUPPORTED_RESBLOCK_KERNELS:
            bad("resblock_kernel_sizes", kernels, f"kernel size {k} has no conv instance (built: "
                f"{SUPPORTED_RESBLOCK_KERNELS})")
    for d in dils:
        for v in d:
            if v not in SUPPORTED_RESBLOCK_DILATIONS:
                bad("resblock_dilation_sizes", dils, f"dilation {v} has no conv instance (built: "
                    f"{SUPPORTED_RESBLOCK_DILATIONS})")
    rates, uk = list(cfg["upsample_rates"]), list(cfg["upsample_kernel_sizes"])
    if len(rates) != len(uk) or not rates:
        bad("upsample_kernel_sizes", uk, f"one kernel size per entry of upsample_rates ({len(rates)}) is required")
    ch = cfg["upsample_initial_channel"]
    for i, (u, k) in enumerate(zip(rates, uk)):
        if k != 2 * u:
            bad("upsample_kernel_sizes", uk, f"stage {i}: the transposed conv is built as a 3-tap phase conv, which needs "
                f"kernel == 2 * stride (got kernel {k}, stride {u})")
        if u <= 0 or 32 % u != 0:
            bad("upsample_rates", rates, f"stage {i}: stride {u} must divide 32 (phases are interleaved inside one "
                f"32-row fragment)")
        if ch % 2 != 0 or (ch // 2) % 32 != 0:
            bad("upsample_initial_channel", cfg["upsample_initial_channel"], f"stage {i} would have {ch // 2} channels; "
                f"every stage needs a multiple of 32 (whole MFMA fragments)")
        ch //= 2
    if cfg["hidden_channels"] % 32 != 0:
        bad("hidden_channels", cfg["hidden_channels"], "must be a multiple of 32 (the WaveNet gate rows are paired in "
            "32-row MFMA fragments)")
    if cfg["inter_channels"] % 64 != 0:
        bad("inter_channels", cfg["inter_channels"], "must be a multiple of 64 (the couplings split the channels in halves "
            "of whole 32-row fragments)")


def generator_margin_frames(cfg, conv_pre_kernel=7, conv_post_kernel=7):
    """Frames beyond an utterance's end that can still influence its last sample: an UPPER BOUND of the generator's
    one-sided receptive field from the configuration (reference: openvoice/models.py:225-291 -- conv_pre, per stage a
    ConvTranspose1d of kernel k_u and stride u followed by the MRF whose widest ResBlock1 reaches
    (k - 1) / 2 * (sum(dilations) + len(dilations)) samples (modules.py:221-309: one dilated and one plain conv per
    dilation), conv_post), rounded up plus one frame of slack.  The transposed-conv term charges (k_u - u + 1) / spf
    frames per stage, more than the ceil((k_u - u) / 2 / u) + 1 inputs a stage really reaches back: the bound is loose
    by design.  15 for the released V1 / V2 configurations (``ConverterEngine`` uses max(GENERATOR_MARGIN = 16, this):
    tests/test_host_algebra_cpu.py::test_generator_margin_released_configs)."""
    import math
    reach = (conv_pre_kernel - 1) / 2.0                      # frames
    spf = 1                                                  # samples per frame after the stage
    for u, ku in zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"]):
        spf *= u
        reach += (ku - u + 1) / 2.0 / spf * 2                # a transposed conv reaches ceil((k_u - u) / 2) + 1 inputs back
        widest = max((k - 1) // 2 * (sum(d) + len(d)) for k, d in
                     zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]))
        reach += widest / spf
    reach += (conv_post_kernel - 1) / 2.0 / spf
    return int(math.ceil(reach)) + 1


def on_own_device(method):
    """Run an engine method with the engine's device current: kernels go to ``current_stream(self.device)`` already,
    but event records (profile mode), HIP-graph capture and the per-device launch sizing inside the library follow
    the CURRENT device, which need not be the engine's (``ToneColorConverter(cfg, device='cuda:1')`` while cuda:0 is
    current)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        dev = getattr(self, "device", None)
        if dev is None:      # GraphedConversion: the engine is an attribute, or the first constructor argument
            dev = (getattr(self, "engine", None) or kwargs.get("engine") or args[0]).device
        with torch.cuda.device(dev):
            return method(self, *args, **kwargs)
    return wrapper


def _ptr(t, offset_elems=0):
    return ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


class PackedConv:
    """One conv layer in kernel-ready form: fragment-packed weights + bias in packed row order."""

    def __init__(self, w_dense, bias, device, K, dil=1, cout=None):
        w_dense = w_dense.detach().to(torch.float32).cpu().contiguous()
        rows, cin, k = w_dense.shape
        assert k == K
        self.rows = rows                        # meaningful packed rows (M of the launch)
        self.cout = rows if cout is None else cout
        self.cin, self.K, self.dil = cin, K, dil
        n = _lib.call("ov_conv1d_pack_size", rows, cin, K)
        packed = torch.empty(n, dtype=torch.float32)
        _lib.call("ov_conv1d_pack_f32", w_dense, rows, cin, K, packed)
        self.w = packed.to(device)
        pack_rows = _lib.call("ov_conv1d_pack_rows", rows)
        b = torch.zeros(pack_rows, dtype=torch.float32)
        if bias is not None:
            b[:rows] = bias.detach().float().cpu()
        self.has_bias = bias is not None
        self.bias = b.to(device)


def padded_frames(T):
    """Row stride of the frame-rate tensors inside the engine: T rounded up to 4 floats so that every
    row starts 16-byte aligned and the conv kernels take their 16-byte staging path for any T."""
    return (T + 3) // 4 * 4


def gate_row_order(hidden):
    """Packed row order pairing row c (tanh | m) with row c+hidden (sigmoid | logs) in adjacent
    32-row MFMA tiles, so both halves of a gate land in the same lane of one wave."""
    assert hidden % 32 == 0
    idx = []
    for q in range(hidden // 32):
        idx += list(range(32 * q, 32 * q + 32)) + list(range(hidden + 32 * q, hidden + 32 * q + 32))
    return torch.tensor(idx, dtype=torch.long)


def conv_transpose_as_conv(w, stride):
    """ConvTranspose1d(k = 2*stride, padding = stride/2) as a 3-tap stride-1 conv producing
    ``stride`` output phases per input frame.  y[s*q + p] = sum_i x[i] * w[:, :, s*q + p + pad - s*i]
    has exactly two non-zero taps per phase (i = q and i = q-1 or q+1); row = cout*stride + phase.
    ``w`` is [Cin, Cout, k] (torch ConvTranspose1d layout, reference: openvoice/models.py:244-256)."""
    cin, cout, k = w.shape
    s = stride
    pad = (k - s) // 2
    assert k == 2 * s and (k - s) % 2 == 0
    wc = torch.zeros(cout * s, cin, 3, dtype=w.dtype)
    for p in range(s):
        rows = torch.arange(cout) * s + p
        wc[rows, :, 1] = w[:, :, p + pad].t()                 # x[q]
        if p + pad < s:
            wc[rows, :, 0] = w[:, :, p + pad + s].t()         # x[q-1]
        else:
            wc[rows, :, 2] = w[:, :, p + pad - s].t()         # x[q+1]
    return wc


def convt_row_order(cout, stride):
    """Packed row order of the grouped ConvTranspose kernels (``F_CONVT_GROUPED``, include/openvoice_amd.h
    OV_EPI_CONVT): natural row ``cout*stride + phase`` of ``conv_transpose_as_conv`` -> 32-row tiles that hold ONE
    phase group each -- tile 2q the phases < stride/2 (their tap x[t+1] is all zeros), tile 2q+1 the phases >=
    stride/2 (tap x[t-1] all zeros) of the same output channels -- so the kernel skips a third of the matrix
    work.  Returns None when the shape cannot be grouped (the natural order + generic kernel are used then)."""
    s = stride
    if s == 8 and cout % 8 == 0:
        per_tile = 8          # channels per tile pair; row-in-tile = 4 * (cout % 8) + phase % 4
    elif s == 2 and cout % 32 == 0:
        per_tile = 32         # row-in-tile = cout % 32
    else:
        return None
    idx = []
    for q in range(cout // per_tile):
        for grp in range(2):
            for c in range(per_tile):
                for ph in range(s // 2):
                    idx.append((q * per_tile + c) * s + grp * (s // 2) + ph)
    return torch.tensor(idx, dtype=torch.long)


# measurement knob (A/B runs of the same process image): flag bits OR-ed into every conv launch, e.g.
# OPENVOICE_AMD_CONV_FLAGS=8 (OV_F_NO_XCD_MAP) restores the round-robin tile order
_EXTRA_CONV_FLAGS = int(os.environ.get("OPENVOICE_AMD_CONV_FLAGS", "0"))
if _EXTRA_CONV_FLAGS & ~_lib.F_NO_XCD_MAP:
    # every other bit (MASK_V, OUT2_INIT, CONVT_GROUPED) changes RESULTS; an environment variable must not do that
    raise _lib.OvError(f"OPENVOICE_AMD_CONV_FLAGS={_EXTRA_CONV_FLAGS}: only the measurement bit OV_F_NO_XCD_MAP "
                       f"({_lib.F_NO_XCD_MAP}) may be set from the environment")


def launch_conv(layer, x, x_off, x_bs, out, out_off, out_bs, B, L, epi=EPI_LINEAR, flags=0, in_slope=1.0,
                scale=1.0, res=None, res_off=0, res_bs=0, add=None, add_bs=0, out2=None, out2_bs=0, mask=None,
                bias_b=None, bias_b_off=0, bias_b_bs=0, split=0, phase_s=1, cin=None, rows=None, tiles_per_wg=0,
                x_ld=0, out_ld=0, mask_bs=0, tile=0, loaders=0, chunk=0, col_limit=None, col_limit_scale=1):
    """Fill ``ov_conv1d_params`` and launch on torch's current stream of ``x``'s device.
    Offsets, row strides (``*_ld``, 0 = dense) and batch strides are in elements.  ``col_limit`` (int32 [B] on the
    device) x ``col_limit_scale`` = output columns per utterance that matter: tiles beyond are not computed."""
    flags |= _EXTRA_CONV_FLAGS
    if _lib.use_torch_binding():
        ip = [B, layer.cin if cin is None else cin, L, x_ld, out_ld, layer.rows if rows is None else rows, layer.cout,
              layer.K, layer.dil, epi, flags, split, phase_s, tiles_per_wg, tile, loaders, chunk,
              x_bs, out_bs, res_bs, add_bs, out2_bs, bias_b_bs, mask_bs, x_off, out_off, res_off, bias_b_off,
              col_limit_scale]
        _lib.torch_op("conv1d_f32", x, layer.w, layer.bias, out, res, add, out2, mask, bias_b, col_limit, ip,
                      [in_slope, scale])
        return
    p = ConvParams()
    p.x, p.w = _ptr(x, x_off), _ptr(layer.w)
    p.bias = _ptr(layer.bias)        # zeros when the layer has no bias (the kernel always adds it)
    p.bias_b = _ptr(bias_b, bias_b_off) if bias_b is not None else None
    p.out = _ptr(out, out_off)
    p.res = _ptr(res, res_off) if res is not None else None
    p.add = _ptr(add) if add is not None else None
    p.out2 = _ptr(out2) if out2 is not None else None
    p.mask = _ptr(mask) if mask is not None else None
    p.x_bstride, p.out_bstride, p.res_bstride = x_bs, out_bs, res_bs
    p.add_bstride, p.out2_bstride, p.bias_b_bstride = add_bs, out2_bs, bias_b_bs
    p.B, p.Cin, p.L = B, layer.cin if cin is None else cin, L
    p.M = layer.rows if rows is None else rows
    p.Cout = layer.cout
    p.K, p.dil, p.epi, p.flags, p.split, p.phase_s = layer.K, layer.dil, epi, flags, split, phase_s
    p.in_slope, p.scale = in_slope, scale
    p.tiles_per_wg, p.tile, p.loaders, p.chunk = tiles_per_wg, tile, loaders, chunk
    p.x_ld, p.out_ld, p.mask_bstride = x_ld, out_ld, mask_bs
    p.col_limit = ctypes.c_void_p(col_limit.data_ptr()) if col_limit is not None else None
    p.col_limit_scale = col_limit_scale
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_f32(ctypes.byref(p), stream), "ov_conv1d_f32")


def launch_pair(c1, c2, x, x_bs, out, out_bs, B, L, add=None, add_bs=0, scale=1.0, slope=LRELU_SLOPE, ld=0, nwg=0,
                dbg=None, col_limit=None, col_limit_scale=1):
    """One fused ResBlock1 iteration (``ov_resblock_pair_f32``): out = (c2(lrelu(c1(lrelu(x)))) + x [+ add]) * scale.
    ``c1`` / ``c2`` are the ``PackedConv`` layers of the two convs; ``out`` must not alias ``x``."""
    if _lib.use_torch_binding():
        _lib.torch_op("resblock_pair_f32", x, c1.w, c1.bias, c2.w, c2.bias, out, add, dbg, col_limit,
                      [B, c1.cin, L, ld, c1.K, c1.dil, nwg, x_bs, out_bs, add_bs, col_limit_scale], [slope, scale])
        return
    p = _lib.RespairParams()
    p.x, p.w1, p.b1, p.w2, p.b2 = _ptr(x), _ptr(c1.w), _ptr(c1.bias), _ptr(c2.w), _ptr(c2.bias)
    p.out = _ptr(out)
    p.add = _ptr(add) if add is not None else None
    p.x_bstride, p.out_bstride, p.add_bstride = x_bs, out_bs, add_bs
    p.B, p.C, p.L, p.ld, p.K, p.dil, p.nwg = B, c1.cin, L, ld, c1.K, c1.dil, nwg
    p.slope, p.scale = slope, scale
    p.dbg = ctypes.c_void_p(dbg.data_ptr()) if dbg is not None else None
    p.col_limit = ctypes.c_void_p(col_limit.data_ptr()) if col_limit is not None else None
    p.col_limit_scale = col_limit_scale
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_resblock_pair_f32(ctypes.byref(p), stream), "ov_resblock_pair_f32")


_pair_supported = {}


def pair_supported(C, K, dil):
    key = (C, K, dil)
    if key not in _pair_supported:
        _pair_supported[key] = bool(_lib.call("ov_resblock_pair_supported", C, K, dil))
    return _pair_supported[key]


# Where the fused pair beats its two launches on MI355X (profiles/r02_s5_pair_vs_two_launches.txt, B = 32 x 10 s):
# C = 32, k = 3: 0.91 vs 1.09 ms; C = 32, k = 7: 1.75 vs 1.85; C = 64, k = 3: 1.60 vs 1.64.  The MFMA-bound shapes
# (C = 32 k = 11, C = 64 k = 7) run 4-9 % SLOWER fused (32 accumulator registers per wave instead of 64: less matrix
# work per LDS operand) and stay on the two-launch path.  Round 6: at C = 64, k = 3 two Winograd-domain launches (0.56 +
# 0.73 ms, profiles/r06_s17) beat the fused direct pair (1.62 ms), so that shape left the policy.
PAIR_POLICY = {(32, 3), (32, 7)}


def wino_policy(C, K, dil):
    """Where a Winograd-domain launch (csrc/conv1d_wino.h) beats the direct conv it replaces (profiles/r06_s17, B = 32 x
    10 s): every instance at C >= 64; at C = 32 (K = 11, one 32-row fragment per workgroup: the input transform is shared by
    a single row fragment and its helper waves set the pace) the dilation-1 convs only (1.03 vs 1.21 ms; the dilated
    instances tie with the direct kernel and lose in the residual form)."""
    return C >= 64 or dil == 1


# A Winograd launch deals (utterance, column block, row block) items to ONE workgroup per CU; below about half of the
# chip's CUs in items the direct kernel's small tiles fill the machine better (profiles/r06_s20, r06_s21: generator stage 0,
# C = 256, 6 888 columns -- batch 1: 54 items, Winograd 0.45-0.6x the direct conv; batch 2: 108 items, 0.8-1.0x; batch 3:
# 162 items, ahead; batch 4: 216 items, 1.4-1.7x as at batch 32; every other stage has >= 216 items at batch 1 and runs
# 1.1-1.65x faster).  Batch 1 with the rule: 7.50 -> 6.18 ms.
WINO_MIN_ITEMS = 136

# Opt-in split-precision path (use_split_bf16x3): generator stages with at least this many channels run on ov_conv1d_split3
SPLIT_MIN_CHANNELS = 128


def wino_items(cout, dil, B, L):
    """Work items of one ov_conv1d_wino_f32 launch (csrc/conv1d_wino.h: 256 columns per matrix wave at dilation 1,
    4 * (64 // dil) * dil at dilation dil; 4 / MW column sub-blocks per workgroup)."""
    mw = 4 if cout % 128 == 0 else (2 if cout % 64 == 0 else 1)
    ncol = (256 if dil == 1 else 4 * (64 // dil) * dil) * (4 // mw)
    return B * ((L + ncol - 1) // ncol) * (cout // (32 * mw))


def wn_fused_row_order(hidden):
    """Gate row order of the fused WaveNet-layer kernel (``ov_wn_layer_f32``, include/openvoice_amd.h): 16-row block
    q holds, at rows 4j + {0, 1, 2, 3}, the tanh rows of channels 8q + j and 8q + j + 4, then their sigmoid rows, so
    that the four accumulator rows of a lane of the 16x16x4 MFMA are both halves of two gates."""
    assert hidden % 8 == 0
    idx = []
    for q in range(hidden // 8):
        for j in range(4):
            a, b = 8 * q + j, 8 * q + j + 4
            idx += [a, b, hidden + a, hidden + b]
    return torch.tensor(idx, dtype=torch.long)


def wn_pack(w_dense, device):
    """Dense [rows][cin][K] -> the 16x16x4 fragment order of ``ov_wn_pack_f32`` (device tensor)."""
    w_dense = w_dense.detach().to(torch.float32).cpu().contiguous()
    rows, cin, k = w_dense.shape
    n = _lib.call("ov_wn_pack_size", rows, cin, k)
    assert n > 0, (rows, cin, k)
    packed = torch.empty(n, dtype=torch.float32)
    _lib.call("ov_wn_pack_f32", w_dense, rows, cin, k, packed)
    return packed.to(device)


def launch_wn_layer(layer, x, out, skip, mask, B, T, ld, cond=None, cond_off=0, cond_bs=0, first=False, last=False,
                    width=0, mask_bs=0, dbg=None, acts=None, row_split=0):
    """One fused WaveNet layer (``ov_wn_layer_f32``): out = (x + res) * mask, skip (+)= rs; ``layer`` is a dict of
    the packed tensors built by ``_WaveNet``; x / out / skip are [B][H][ld].  ``acts`` ([B][H][ld] scratch) lets the
    launcher run one or two utterances as the row-split launch pair (bit-identical results); ``row_split`` 1 / 3 force
    the fused / the split form."""
    if _lib.use_torch_binding():
        H = layer["hidden"]
        _lib.torch_op("wn_layer_f32", x, out, skip, layer["w_in"], layer["b_in"], cond, layer["w_rs"], layer["b_rs"],
                      mask, dbg, acts, [B, H, T, ld, layer["K"], int(first), int(last), width, H * ld, cond_bs, mask_bs,
                                        cond_off, row_split])
        return
    p = _lib.WnLayerParams()
    p.x, p.out, p.skip = _ptr(x), _ptr(out), _ptr(skip)
    p.w_in, p.b_in, p.w_rs, p.b_rs = _ptr(layer["w_in"]), _ptr(layer["b_in"]), _ptr(layer["w_rs"]), _ptr(layer["b_rs"])
    p.cond = _ptr(cond, cond_off) if cond is not None else None
    p.mask = _ptr(mask)
    H = layer["hidden"]
    p.bstride, p.cond_bstride, p.mask_bstride = H * ld, cond_bs, mask_bs
    p.B, p.H, p.T, p.ld, p.K = B, H, T, ld, layer["K"]
    p.first, p.last, p.width, p.row_split = int(first), int(last), width, row_split
    p.dbg = ctypes.c_void_p(dbg.data_ptr()) if dbg is not None else None
    p.acts = _ptr(acts) if acts is not None else None
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_wn_layer_f32(ctypes.byref(p), stream), "ov_wn_layer_f32")


class _WaveNet:
    """Packed WN stack (reference: openvoice/modules.py:133-210).  Two forms of every layer: the fused one-launch
    layer (``ov_wn_layer_f32``) where the shape has an instance, and the gate + res/skip launch pair."""

    def __init__(self, sd, prefix, n_layers, hidden, device):
        order = gate_row_order(hidden)
        self.hidden, self.n_layers = hidden, n_layers
        self.in_layers, self.rs_layers, self.fused_layers = [], [], []
        K = sd[f"{prefix}.in_layers.0.weight_v"].shape[2] if f"{prefix}.in_layers.0.weight_v" in sd else \
            sd[f"{prefix}.in_layers.0.weight"].shape[2]
        self.fused = bool(_lib.call("ov_wn_layer_supported", hidden, K))
        forder = wn_fused_row_order(hidden) if self.fused else None
        for i in range(n_layers):
            w_in = effective_weight(sd, f"{prefix}.in_layers.{i}")
            b_in = sd[f"{prefix}.in_layers.{i}.bias"].float()
            self.in_layers.append(PackedConv(w_in[order], b_in[order], device, K=w_in.shape[2], cout=hidden))
            w_rs = effective_weight(sd, f"{prefix}.res_skip_layers.{i}")
            b_rs = sd[f"{prefix}.res_skip_layers.{i}.bias"].float()
            self.rs_layers.append(PackedConv(w_rs, b_rs, device, K=1))
            if self.fused:
                if w_rs.shape[0] == hidden:      # last layer: skip rows only (modules.py:170-173, :203-207)
                    w_rs = torch.cat([torch.zeros_like(w_rs), w_rs])
                    b_rs = torch.cat([torch.zeros_like(b_rs), b_rs])
                self.fused_layers.append(dict(hidden=hidden, K=w_in.shape[2], w_in=wn_pack(w_in[forder], device),
                                              b_in=b_in[forder].contiguous().to(device), w_rs=wn_pack(w_rs, device),
                                              b_rs=b_rs.contiguous().to(device)))
        # cond_layer rows re-ordered per layer so its output is directly the gate's per-batch bias
        wc = effective_weight(sd, prefix + ".cond_layer")[:, :, 0]
        bc = sd[prefix + ".cond_layer.bias"].float()
        full = torch.cat([order + 2 * hidden * i for i in range(n_layers)])
        self.cond_w = wc[full].contiguous().to(device)
        self.cond_b = bc[full].contiguous().to(device)
        if self.fused:
            full = torch.cat([forder + 2 * hidden * i for i in range(n_layers)])
            self.cond_w_fused = wc[full].contiguous().to(device)
            self.cond_b_fused = bc[full].contiguous().to(device)


class ConverterEngine:
    """Kernel-level implementation of the converter model for one device."""

    def __init__(self, state_dict, model_cfg, spec_channels, device, zero_g=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.OvError("ConverterEngine needs a ROCm device ('cuda:N'); there is no CPU path")
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        cfg = dict(model_cfg.items()) if hasattr(model_cfg, "items") else dict(model_cfg)
        validate_config(cfg)          # names the unsupported field before any weight is packed
        self.cfg = cfg
        self.zero_g = bool(zero_g)
        self.inter = cfg["inter_channels"]
        self.hidden = cfg["hidden_channels"]
        self.gin = cfg.get("gin_channels", 256)
        self.spec_channels = spec_channels
        dev = self.device
        H, C = self.hidden, self.inter
        half = C // 2
        # ---- posterior encoder -----------------------------------------------------------------
        self.q_pre = PackedConv(sd["enc_q.pre.weight"], sd["enc_q.pre.bias"], dev, K=1)
        self.q_wn = _WaveNet(sd, "enc_q.enc", ENC_Q_LAYERS, H, dev)
        order = gate_row_order(C)
        self.q_proj = PackedConv(sd["enc_q.proj.weight"][order], sd["enc_q.proj.bias"][order], dev, K=1, cout=C)
        # ---- flow: Flip folded into channel order of pre (inputs) / post (outputs) ---------------
        self.couplings = []
        for f in range(N_FLOWS):
            p = f"flow.flows.{2 * f}"
            flipped = f % 2 == 1   # an odd number of Flips precedes this coupling in both directions
            w_pre, w_post, b_post = sd[p + ".pre.weight"], sd[p + ".post.weight"], sd[p + ".post.bias"]
            if flipped:
                w_pre = torch.flip(w_pre, [1])
                w_post, b_post = torch.flip(w_post, [0]), torch.flip(b_post, [0])
            self.couplings.append(dict(
                flipped=flipped,
                pre=PackedConv(w_pre, sd[p + ".pre.bias"], dev, K=1),
                wn=_WaveNet(sd, p + ".enc", FLOW_LAYERS, H, dev),
                post=PackedConv(w_post, b_post, dev, K=1)))
        self.half = half
        # ---- generator -------------------------------------------------------------------------
        self.conv_pre = PackedConv(sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], dev, K=7)
        self.dec_cond_w = sd["dec.cond.weight"][:, :, 0].contiguous().to(dev)
        self.dec_cond_b = sd["dec.cond.bias"].contiguous().to(dev)
        self.ups, self.resblocks = [], []
        ch = cfg["upsample_initial_channel"]
        kernels, dils = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            w = effective_weight(sd, f"dec.ups.{i}")
            wc = conv_transpose_as_conv(w, u)
            bias = sd[f"dec.ups.{i}.bias"].repeat_interleave(u)
            order = convt_row_order(ch // 2, u)
            if order is not None:
                wc, bias = wc[order], bias[order]
            self.ups.append(dict(conv=PackedConv(wc, bias, dev, K=3, cout=ch // 2), stride=u,
                                 flags=F_CONVT_GROUPED if order is not None else 0))
            ch //= 2
            stage = []
            for j, (rk, rd) in enumerate(zip(kernels, dils)):
                rb = f"dec.resblocks.{i * len(kernels) + j}"
                pairs = []
                for n, d in enumerate(rd):
                    c1 = PackedConv(effective_weight(sd, f"{rb}.convs1.{n}"), sd[f"{rb}.convs1.{n}.bias"], dev, K=rk, dil=d)
                    c2 = PackedConv(effective_weight(sd, f"{rb}.convs2.{n}"), sd[f"{rb}.convs2.{n}.bias"], dev, K=rk, dil=1)
                    pairs.append((c1, c2))
                stage.append(pairs)
            self.resblocks.append(stage)
        # Winograd-domain twins (csrc/conv1d_wino.h) of the ResBlock convs of the MFMA-bound stages: the same fp32 conv
        # with 1.6-2x fewer executed multiplies; per stage / ResBlock / pair (w1 | None, w2 | None)
        self.wino_resblocks = []
        from . import wino as _wino
        ch = cfg["upsample_initial_channel"]
        for i in range(len(cfg["upsample_rates"])):
            ch //= 2
            stage = []
            for j, (rk, rd) in enumerate(zip(kernels, dils)):
                rb = f"dec.resblocks.{i * len(kernels) + j}"
                pairs = []
                for n, d in enumerate(rd):
                    w1 = w2 = None
                    if _wino.supported(ch, ch, rk, d) and wino_policy(ch, rk, d):
                        w1 = _wino.PackedConvWino(effective_weight(sd, f"{rb}.convs1.{n}"), sd[f"{rb}.convs1.{n}.bias"], dev, dil=d)
                    if _wino.supported(ch, ch, rk, 1) and wino_policy(ch, rk, 1):
                        w2 = _wino.PackedConvWino(effective_weight(sd, f"{rb}.convs2.{n}"), sd[f"{rb}.convs2.{n}.bias"], dev, dil=1)
                    pairs.append((w1, w2))
                stage.append(pairs)
            self.wino_resblocks.append(stage)
        self.final_channels = ch
        self.post_w = sd["dec.conv_post.weight"][0].contiguous().to(dev)   # [C, 7]
        self.total_upsample = 1
        for u in cfg["upsample_rates"]:
            self.total_upsample *= u
        # ---- reference encoder (extract_se): present in converter checkpoints only -------------------
        self.ref_enc = None
        if "ref_enc.gru.weight_ih_l0" in sd:
            convs = [(effective_weight(sd, f"ref_enc.convs.{i}").contiguous().to(dev),
                      sd[f"ref_enc.convs.{i}.bias"].contiguous().to(dev)) for i in range(len(REF_ENC_FILTERS))]
            w_ih = sd["ref_enc.gru.weight_ih_l0"]
            self.ref_enc = dict(
                ln_w=sd["ref_enc.layernorm.weight"].contiguous().to(dev),
                ln_b=sd["ref_enc.layernorm.bias"].contiguous().to(dev),
                convs=convs,
                gru_in=PackedConv(w_ih.unsqueeze(-1), sd["ref_enc.gru.bias_ih_l0"], dev, K=1),
                whh_t=sd["ref_enc.gru.weight_hh_l0"].t().contiguous().to(dev),
                bhh=sd["ref_enc.gru.bias_hh_l0"].contiguous().to(dev),
                proj_w=sd["ref_enc.proj.weight"].contiguous().to(dev),
                proj_b=sd["ref_enc.proj.bias"].contiguous().to(dev))
        self._ws = {}
        self.profile = None   # set to [] to collect per-launch HIP-event timings
        # independent ResBlock chains of a generator stage on this many HIP streams when the batch is at most
        # chain_streams_max_batch utterances (decode()); 1 = always the serial one-stream order
        # Measured (round 4, profiles/r04_s14_sweep_chain_ab.txt): batch 1 8.75 vs 8.81 ms, batch 4 20.4 vs 20.2 ms -- the
        # chains do overlap, but every kernel then runs proportionally longer (the chip is occupied by one-tile
        # workgroups either way) -- so the default is the serial order; results are bit-identical both ways.
        self.chain_streams = 1
        self.chain_streams_max_batch = 8
        self._streams = []
        self._zeros = {}
        # frames the generator computes beyond an utterance's length under skip_padding: this checkpoint's own
        # receptive field (larger ResBlock kernels / dilations or other upsampling than the released models' would
        # silently corrupt the last samples with a hard-wired 16)
        self.generator_margin = max(GENERATOR_MARGIN, generator_margin_frames(self.cfg))
        self.fuse_wn = True      # WaveNet layers as one launch each (ov_wn_layer_f32) where the shape has an instance
        self.wn_row_split = 0    # 0: the launcher splits a layer's rows over two launches for one or two utterances; 1: never
        self.fuse_pairs = True   # ResBlock pairs of the HBM-bound stages as one launch each (PAIR_POLICY)
        # ResBlock convs in the Winograd domain where wino_policy() says so (ov_conv1d_wino_f32):
        # fp32 arithmetic, 6 G / (4 K) of the direct form's multiplies, ~4x its rounding error
        self.use_winograd = True
        # opt-in fast generator: bf16 activations, fp32 accumulation (waveform within
        # ~1e-2 of the fp32 path instead of ~1e-5).  Built lazily by use_bf16_generator().
        self._state_dict_for_bf16 = sd
        self.generator_bf16 = None
        # opt-in split-precision MRF stages: fp32-level products on the bf16 matrix pipe
        self._split3_on = False
        self.split3_products = 6
        self.split_resblocks = None      # per stage: None, or [[(c1, c2) PackedConvSplit3 pairs] per ResBlock]

    # ---- launch helpers --------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _conv(self, layer, x, x_off, x_bs, out, out_off, out_bs, B, L, tag="conv", alg_flops=None, **kwargs):
        """Launch one conv; when ``self.profile`` is a list, bracket the launch with HIP events on
        the launch stream and record (tag, algorithmic FLOPs, start, end) for bench.py's roofline."""
        if self.profile is None:
            launch_conv(layer, x, x_off, x_bs, out, out_off, out_bs, B, L, **kwargs)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_conv(layer, x, x_off, x_bs, out, out_off, out_bs, B, L, **kwargs)
        e1.record()
        if alg_flops is None:
            alg_flops = 2.0 * layer.rows * (kwargs.get("cin") or layer.cin) * layer.K * L * B
        self.profile.append((tag, alg_flops, e0, e1))

    def _wino(self, layer, x, out, bs, B, L, res=None, add=None, scale=1.0, in_slope=LRELU_SLOPE, out_slope=1.0, **lim):
        """One Winograd-domain ResBlock conv (leaky ReLU on the input, LINEAR epilogue); profiled under the MRF tag with its
        algorithmic FLOPs and, as a fifth field, the FLOPs the kernel EXECUTES (wino.PRODUCTS_PER_TILE[K] / (4 K) of them)."""
        from . import wino
        kw = dict(in_slope=in_slope, out_slope=out_slope, scale=scale, res=res, res_bs=bs if res is not None else 0, add=add,
                  add_bs=bs if add is not None else 0, **lim)
        if self.profile is None:
            wino.launch_conv_wino(layer, x, bs, out, bs, B, L, **kw)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wino.launch_conv_wino(layer, x, bs, out, bs, B, L, **kw)
        e1.record()
        alg = 2.0 * layer.cout * layer.cin * layer.K * L * B
        self.profile.append(("mrf", alg, e0, e1, alg * wino.PRODUCTS_PER_TILE[layer.K] / (4.0 * layer.K)))

    def _pair(self, c1, c2, x, out, bs, B, L, add, scale, **lim):
        """One fused ResBlock1 iteration; profiled under the same tag as the two launches it replaces, with their
        algorithmic FLOPs (both convs)."""
        if self.profile is None:
            launch_pair(c1, c2, x, bs, out, bs, B, L, add=add, add_bs=bs, scale=scale, **lim)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_pair(c1, c2, x, bs, out, bs, B, L, add=add, add_bs=bs, scale=scale, **lim)
        e1.record()
        self.profile.append(("mrf", 2 * 2.0 * c1.rows * c1.cin * c1.K * L * B, e0, e1))

    def _linear(self, x2d, w, b):
        Bg, Kd = x2d.shape
        M = w.shape[0]
        y = torch.empty(Bg, M, dtype=torch.float32, device=self.device)
        _lib.call("ov_linear_f32", x2d, w, b, y, Bg, M, Kd)
        return y

    def _wn_cond(self, wn, g):
        """cond_layer (modules.py:189-190) for every layer of ``wn`` at once, rows in the order the gate kernel in
        use expects them as its per-utterance bias."""
        if self.fuse_wn and wn.fused:
            return self._linear(g, wn.cond_w_fused, wn.cond_b_fused)
        return self._linear(g, wn.cond_w, wn.cond_b)

    def _workspace(self, B, T):
        key = (B, T)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()   # one resident shape at a time; the decoder scratch is GBs at B=32
            dev, H, C = self.device, self.hidden, self.inter
            Tp = padded_frames(T)
            # zero-filled once: the pad columns [T, Tp) are never written by a kernel and never read as data
            f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
            lat = f(3, B, C, Tp)         # the three latents in ONE allocation: one ov_unpad_rows_f32 returns them dense
            ws = dict(Tp=Tp, mask=f(B, Tp), h=f(B, H, Tp), h2=f(B, H, Tp), acts=f(B, H, Tp), skip=f(B, H, Tp), noise=f(B, C, Tp),
                      lat=lat, z=lat[0], z_p=lat[1], z_hat=lat[2])
            ch = self.cfg["upsample_initial_channel"]
            ws["pre"] = f(B, ch, Tp)
            f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            L, biggest = T, 0
            for u in self.cfg["upsample_rates"]:
                ch //= 2
                L *= u
                biggest = max(biggest, ch * L)
            ws["dec"] = [f(B * biggest) for _ in range(5)]
            self._ws[key] = ws
        return ws

    def _zeros_like_cached(self, t):
        """An all-zero tensor of ``t``'s shape (the ``zero_g`` conditioning, models.py:495,498) without a fill launch
        per conversion."""
        key = tuple(t.shape)
        z = self._zeros.get(key)
        if z is None:
            z = self._zeros[key] = torch.zeros_like(t)
        return z

    def _side_streams(self, n):
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(self.device))
        return self._streams[:n]

    def _chain_scratch(self, ws, n):
        """``n`` more decoder scratch buffers (the concurrent chains' own t1 / ra), allocated on first use."""
        extra = ws.setdefault("dec_extra", [])
        while len(extra) < n:
            extra.append(torch.empty_like(ws["dec"][0]))
        return extra

    def _wavenet(self, wn, ws, B, T, cond, mask):
        """h (in place) -> skip accumulator; reference: openvoice/modules.py:192-210.  The final
        ``output * x_mask`` (modules.py:210) is dropped: every consumer (proj / post) is a 1x1 conv
        whose own epilogue multiplies by the same 0/1 mask, which makes it a no-op."""
        H, Tp = wn.hidden, ws["Tp"]
        cbs = 0 if cond.shape[0] == 1 else cond.shape[1]
        if self.fuse_wn and wn.fused:
            # one launch per layer; h ping-pongs between two buffers (a tile reads its neighbours' halo columns)
            src, dst = ws["h"], ws["h2"]
            for i in range(wn.n_layers):
                layer = wn.fused_layers[i]
                args = dict(cond=cond, cond_off=2 * H * i, cond_bs=cbs, first=i == 0, last=i == wn.n_layers - 1,
                            mask_bs=Tp, acts=ws["acts"], row_split=self.wn_row_split)
                if self.profile is None:
                    launch_wn_layer(layer, src, dst, ws["skip"], mask, B, T, Tp, **args)
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch_wn_layer(layer, src, dst, ws["skip"], mask, B, T, Tp, **args)
                    e1.record()
                    rs_rows = H if args["last"] else 2 * H
                    self.profile.append(("wn_layer", 2.0 * (2 * H * H * layer["K"] + rs_rows * H) * T * B, e0, e1))
                src, dst = dst, src
            return
        for i in range(wn.n_layers):
            self._conv(wn.in_layers[i], ws["h"], 0, H * Tp, ws["acts"], 0, H * Tp, B, T, epi=EPI_GATE,
                       bias_b=cond, bias_b_off=2 * H * i, bias_b_bs=cbs, rows=2 * H, x_ld=Tp, out_ld=Tp, tag="wn_in")
            last = i == wn.n_layers - 1
            self._conv(wn.rs_layers[i], ws["acts"], 0, H * Tp, ws["h"], 0, H * Tp, B, T, epi=EPI_RESSKIP,
                       flags=F_OUT2_INIT if i == 0 else 0, out2=ws["skip"], out2_bs=H * Tp, mask=mask, mask_bs=Tp,
                       split=0 if last else H, x_ld=Tp, out_ld=Tp, tag="wn_rs")

    def _flow(self, src, buf, ws, B, T, conds, mask, reverse):
        """ResidualCouplingBlock (models.py:390-397) from ``src`` into ``buf`` without a copy: a coupling rewrites one
        half of the channels (x1) and leaves the other (x0) alone, and consecutive couplings alternate halves (the
        folded Flips), so the first two couplings read their x1 operand from ``src`` and write it to ``buf`` -- after
        them every channel of ``buf`` has been written -- and the rest run in place."""
        C, half, H, Tp = self.inter, self.half, self.hidden, ws["Tp"]
        order = range(N_FLOWS - 1, -1, -1) if reverse else range(N_FLOWS)
        for n, f in enumerate(order):
            cp = self.couplings[f]
            x0_off = half * Tp if cp["flipped"] else 0
            x1_off = 0 if cp["flipped"] else half * Tp
            self._conv(cp["pre"], src if n == 0 else buf, x0_off, C * Tp, ws["h"], 0, H * Tp, B, T, flags=F_MASK_V,
                       mask=mask, mask_bs=Tp, x_ld=Tp, out_ld=Tp, tag="cpl_pre")
            self._wavenet(cp["wn"], ws, B, T, conds[f], mask)
            self._conv(cp["post"], ws["skip"], 0, H * Tp, buf, x1_off, C * Tp, B, T, epi=EPI_COUPLE, mask=mask,
                       mask_bs=Tp, x_ld=Tp, out_ld=Tp, scale=-1.0 if reverse else 1.0, tag="cpl_post",
                       res=src if n < 2 else None, res_off=x1_off, res_bs=C * Tp)

    # ---- the path ----------------------------------------------------------------------------------
    @torch.no_grad()
    @on_own_device
    def voice_conversion(self, spec, spec_lengths, sid_src, sid_tgt, tau=1.0, noise=None, skip_padding=False):
        """Same contract as the reference seam (openvoice/models.py:492-499):
        ``(o_hat [B,1,256T], y_mask [B,1,T], (z, z_p, z_hat) [B,192,T])``.  ``noise`` [B,192,T]
        replaces the reference's ``torch.randn_like`` draw (models.py:220); when omitted it is drawn
        from torch's generator on the device.  ``skip_padding``: the generator -- 98 % of the work, and unmasked in
        the reference, so a padded batch costs as if every utterance had the longest length -- computes only the
        first ``length + GENERATOR_MARGIN`` frames of each utterance (length-aware work lists,
        ``ov_conv1d_params.col_limit``); every sample of the first ``length`` frames is bit-identical to the full
        computation, and everything beyond them in ``o_hat`` is zero instead of the reference's bias-driven junk."""
        dev = self.device
        spec = spec.to(dev, torch.float32)
        B, F, T = spec.shape
        assert F == self.spec_channels
        # rows may be padded (the native spectrogram returns a [B, F, T] view of 16-byte aligned rows)
        if not (spec.stride(2) == 1 and spec.stride(1) >= T and spec.stride(0) >= F * spec.stride(1)):
            spec = spec.contiguous()
        C = self.inter
        lengths = spec_lengths.to(dev, torch.int64).contiguous()
        g_src = sid_src.to(dev, torch.float32).reshape(sid_src.shape[0], -1).contiguous()
        g_tgt = sid_tgt.to(dev, torch.float32).reshape(sid_tgt.shape[0], -1).contiguous()
        ws = self._workspace(B, T)
        Tp, mask = ws["Tp"], ws["mask"]
        if noise is None:
            # one launch, the RNG stream of a dense torch.randn(B, C, T) (the reference's randn_like, models.py:220):
            # the pad columns [T, Tp) are not drawn into, so a seeded conversion does not depend on the row padding
            ws["noise"][:, :, :T].normal_()
        else:
            ws["noise"][:, :, :T].copy_(noise.to(dev, torch.float32))     # into the padded-row layout
        _lib.call("ov_sequence_mask_f32", lengths, mask, B, T, Tp)
        # conditioning GEMVs (T = 1): modules.py:189-190 for every WN, models.py:275 for the decoder
        g_q = self._zeros_like_cached(g_src) if self.zero_g else g_src
        g_d = self._zeros_like_cached(g_tgt) if self.zero_g else g_tgt
        conds = dict(q=self._wn_cond(self.q_wn, g_q),
                     src=[self._wn_cond(cp["wn"], g_src) for cp in self.couplings],
                     tgt=[self._wn_cond(cp["wn"], g_tgt) for cp in self.couplings])
        cond_d = self._linear(g_d, self.dec_cond_w, self.dec_cond_b)
        # ---- posterior encoder + flows at frame rate ---------------------------------------------------
        # (measured and dropped: issuing this part -- ~100 launches of <= 1.3 tiles per workgroup slot -- as 2 / 4
        # independent sub-batch chains on separate HIP streams, so that one chain's tiles would fill the other's
        # launch tails, takes 16.1 / 14.0 ms instead of 12.1 ms: profiles/r02_s7_frame_streams_ab.txt)
        self._frames(ws, 0, B, spec, conds, tau)
        z, z_p, z_hat = ws["z"], ws["z_p"], ws["z_hat"]
        # ---- generator (models.py:272-291); z_hat * y_mask is the identity (z_hat already masked) --
        if getattr(self, "_bf16_on", False):
            if self.profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            o_hat = self.generator_bf16.decode(z_hat[:, :, :T], g_d.unsqueeze(-1))
            if self.profile is not None:
                e1.record()
                self.profile.append(("gen_bf16", 0.0, e0, e1))
            if skip_padding:
                # the bf16 generator has no length-aware work lists (it computes the padded batch); the contract of
                # skip_padding -- every sample beyond an utterance's own length is zero -- is kept by masking
                keep_samples = (lengths.clamp(max=T) * self.total_upsample).view(B, 1, 1)
                o_hat.masked_fill_(torch.arange(o_hat.shape[2], device=dev).view(1, 1, -1) >= keep_samples, 0.0)
        else:
            limits = self.frame_limits(lengths, T) if skip_padding else None
            o_hat = self.decode(z_hat, cond_d, ws, T=T, limits=limits)
        # fresh dense tensors for the caller (the workspace is reused by the next call): padded rows -> [.., T]
        dense = torch.empty(3, B, C, T, dtype=torch.float32, device=dev)
        _lib.call("ov_unpad_rows_f32", ws["lat"], dense, 3 * B * C, T, Tp)
        y_mask = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
        _lib.call("ov_unpad_rows_f32", mask, y_mask, B, T, Tp)
        return o_hat, y_mask, (dense[0], dense[1], dense[2])

    def _frames(self, ws, b0, b1, spec, conds, tau):
        """Posterior encoder and the two flow passes for the utterances [b0, b1) (views of the whole-batch workspace)."""
        nb, T, Tp = b1 - b0, spec.shape[2], ws["Tp"]
        C, H = self.inter, self.hidden
        sub = {k: (v[b0:b1] if torch.is_tensor(v) else v) for k, v in ws.items() if k != "dec"}
        rows = lambda t: t if t.shape[0] == 1 else t[b0:b1]
        mask = sub["mask"]
        x = spec[b0:b1]
        # ---- posterior encoder (models.py:212-221) -------------------------------------------------
        self._conv(self.q_pre, x, 0, spec.stride(0), sub["h"], 0, H * Tp, nb, T, flags=F_MASK_V, mask=mask, mask_bs=Tp,
                   x_ld=spec.stride(1), out_ld=Tp, tag="q_pre")
        self._wavenet(self.q_wn, sub, nb, T, rows(conds["q"]), mask)
        z, z_p, z_hat = sub["z"], sub["z_p"], sub["z_hat"]
        self._conv(self.q_proj, sub["skip"], 0, H * Tp, z, 0, C * Tp, nb, T, epi=EPI_POSTERIOR, res=sub["noise"],
                   res_bs=C * Tp, scale=float(tau), mask=mask, mask_bs=Tp, rows=2 * C, x_ld=Tp, out_ld=Tp, tag="q_proj")
        # ---- flow forward with g_src, reverse with g_tgt (models.py:496-497) -----------------------
        self._flow(z, z_p, sub, nb, T, [rows(c) for c in conds["src"]], mask, reverse=False)
        self._flow(z_p, z_hat, sub, nb, T, [rows(c) for c in conds["tgt"]], mask, reverse=True)

    def graphed(self, B, T, tau, src_rows=1, tgt_rows=1, max_cached=4, skip_padding=False):
        """``voice_conversion`` for one fixed (B, T, tau) as a captured HIP graph (see ``GraphedConversion``);
        the most recent ``max_cached`` shapes stay resident."""
        key = (int(B), int(T), float(tau), int(src_rows), int(tgt_rows), bool(getattr(self, "_bf16_on", False)),
               bool(skip_padding), bool(self._split3_on), self.split3_products)
        cache = self.__dict__.setdefault("_graphs", {})
        g = cache.pop(key, None)
        if g is None:
            g = GraphedConversion(self, B, T, tau, src_rows, tgt_rows, skip_padding=skip_padding)
            while len(cache) >= max_cached:
                cache.pop(next(iter(cache)))
        cache[key] = g     # re-inserted last: the dict is the LRU order
        return g

    def use_bf16_generator(self, enable=True):
        """Route ``voice_conversion``'s generator through the bf16 kernels (BASELINE.json configs[4]).  Off by
        default: the fp32 path is the one held to the 1e-3 parity bar."""
        if enable and self.generator_bf16 is None:
            from .bf16 import GeneratorBf16
            self.generator_bf16 = GeneratorBf16(self._state_dict_for_bf16, self.cfg, self.device)
        self._bf16_on = bool(enable)
        return self

    def use_split_bf16x3(self, enable=True, products=6):
        """Run the MRF ResBlocks of the generator stages with C >= 128 (``SPLIT_MIN_CHANNELS``; stages 0-1 of the released
        configuration; instances exist down to C = 64) on ``ov_conv1d_split3``: every fp32 operand
        carried as three bf16 planes (lossless), every product as the six plane products of weight >= 2^-18 on the bf16
        matrix pipe with fp32 accumulation -- fp32-level results (against float64 the error is BELOW the fp32 MFMA
        kernels' on every shape, profiles/r05_s2_split3_table_all_shapes.txt) at 1.3-2.1x the DIRECT fp32 kernels' speed.  Since
        round 6 the default path runs these convs in the Winograd domain, and the two are within 1 % of each other: 102.8
        (this path) vs 103.8 ms per batch-32 conversion.  ``products=3`` uses the hi / mid planes only (16-bit operands, ~2e-5 per
        conv, 80 ms).  Off by default: the contract path is the fp32 kernels
        (reference: openvoice/modules.py:296-309, models.py:280-286)."""
        if products not in (6, 3):
            raise _lib.OvError(f"use_split_bf16x3: products must be 6 or 3, got {products!r}")
        if enable and self.split_resblocks is None:
            from . import split3
            sd, cfg = self._state_dict_for_bf16, self.cfg
            kernels, dils = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
            ch = cfg["upsample_initial_channel"]
            stages = []
            for i in range(len(cfg["upsample_rates"])):
                ch //= 2
                ok = all(split3.supported(ch, ch, rk, d) and split3.supported(ch, ch, rk, 1)
                         for rk, rd in zip(kernels, dils) for d in rd)
                # C = 64 (stage 2) stays on the fp32 Winograd launches: its split convs tie with them (20.9 vs 21.5 ms) and
                # cost two layout passes over the stage's tensors (1.3 ms): 104.3 -> 102.8 ms (profiles/r06_s34_*)
                if not ok or ch < SPLIT_MIN_CHANNELS:
                    stages.append(None)
                    continue
                stage = []
                for j, (rk, rd) in enumerate(zip(kernels, dils)):
                    rb = f"dec.resblocks.{i * len(kernels) + j}"
                    stage.append([(split3.PackedConvSplit3(effective_weight(sd, f"{rb}.convs1.{n}"), sd[f"{rb}.convs1.{n}.bias"],
                                                           self.device, dil=d),
                                   split3.PackedConvSplit3(effective_weight(sd, f"{rb}.convs2.{n}"), sd[f"{rb}.convs2.{n}.bias"],
                                                           self.device, dil=1)) for n, d in enumerate(rd)])
                stages.append(stage)
            self.split_resblocks = stages
        self._split3_on = bool(enable)
        self.split3_products = products
        self.__dict__.pop("_graphs", None)        # captured graphs hold the other path's launches
        return self

    def _split_buffers(self, ws, B, ch, L):
        """Plane tensors (3, B, L, ch) bf16 of the split-precision stages: the activated stage input, the conv1 output,
        two ping-pong pair outputs and one result per ResBlock; allocated once per workspace at the largest stage."""
        nk = len(self.cfg["resblock_kernel_sizes"])
        need = 3 * B * L * ch
        bufs = ws.get("split3")
        if bufs is None:
            # sized ONCE per workspace, at the largest split stage of this (batch, frames) -- stage i has ch_i channels at
            # frames x prod(rates[:i + 1]) columns -- and never from a capturing graph's private pool
            if torch.cuda.is_current_stream_capturing():
                raise _lib.OvError("split-precision plane buffers must exist before graph capture: run one eager conversion "
                                   "of this (batch, frames) shape first")
            cfg, frames = self.cfg, L // self._stage_rate(ch)
            c, rate, largest = cfg["upsample_initial_channel"], 1, need
            for i, u in enumerate(cfg["upsample_rates"]):
                c, rate = c // 2, rate * u
                if self.split_resblocks is not None and self.split_resblocks[i] is not None:
                    largest = max(largest, 3 * B * frames * rate * c)
            bufs = ws["split3"] = [torch.empty(largest, dtype=torch.bfloat16, device=self.device) for _ in range(4 + nk)]
        assert bufs[0].numel() >= need
        return [b[:need].view(3, B, L, ch) for b in bufs]

    def _stage_rate(self, ch):
        """Columns per frame at the generator stage that has ``ch`` channels."""
        c, rate = self.cfg["upsample_initial_channel"], 1
        for u in self.cfg["upsample_rates"]:
            c, rate = c // 2, rate * u
            if c == ch:
                return rate
        raise _lib.OvError(f"no generator stage with {ch} channels")

    def _mrf_split(self, stage, u, acc, ws, B, ch, L, limits=None, rate=1):
        """One MRF stage on the split-precision kernels: u (B, ch, L) fp32 raw -> acc (B, ch, L) fp32 = mean of the
        ResBlock1 outputs.  Tensors between the launches are plane tensors stored ACTIVATED (the producer applies the
        next conv's leaky ReLU before it splits); conv2 reads its residual from the pair's activated input and inverts
        the activation in fp32; the last pair of a ResBlock stores raw; the sum / mean is the layout kernel back."""
        from . import split3
        nk = len(stage)
        bufs = self._split_buffers(ws, B, ch, L)
        xa, t, ping, pong, results = bufs[0], bufs[1], bufs[2], bufs[3], bufs[4:]
        prod = self.split3_products
        # length-aware work lists (skip_padding): 128-column tiles beyond an utterance's limit are dropped from every
        # conv's (dense) work list; the two layout kernels stream whole tensors
        lim = dict(col_limit=limits, col_limit_scale=rate) if limits is not None else {}

        def timed(tag, flops, fn):
            if self.profile is None:
                fn()
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            self.profile.append((tag, flops, e0, e1))

        uv = u[:B * ch * L].view(B, ch, L)
        timed("split_layout", 0.0, lambda: split3.to_planes(uv, LRELU_SLOPE, out=xa))
        for j, pairs in enumerate(stage):
            cur = xa
            for n, (c1, c2) in enumerate(pairs):
                last = n == len(pairs) - 1
                flops = 2.0 * ch * ch * c1.K * L * B
                timed("mrf_split", flops, lambda: split3.launch_conv_split3(c1, cur, t, out_slope=LRELU_SLOPE, products=prod, **lim))
                dst = results[j] if last else (pong if cur is ping else ping)
                timed("mrf_split", flops, lambda: split3.launch_conv_split3(c2, t, dst, res=cur, res_slope=LRELU_SLOPE,
                                                                             out_slope=1.0 if last else LRELU_SLOPE,
                                                                             products=prod, **lim))
                cur = dst
        ops = list(results[:nk]) + [None] * (3 - nk)
        if nk > 3:
            raise _lib.OvError("split-precision MRF: at most 3 ResBlocks per stage")
        av = acc[:B * ch * L].view(B, ch, L)
        timed("split_layout", 0.0, lambda: split3.from_planes(ops[0], ops[1], ops[2], in_slope=1.0, scale=1.0 / nk, out=av))

    def frame_limits(self, lengths, T):
        """Two int32 [B] device vectors (``ov_frame_limits_i32``, no host sync): the frames of each utterance the
        generator must COMPUTE so that its first ``lengths[b]`` frames are unaffected by what lies beyond
        (``length + GENERATOR_MARGIN``), and the frames it KEEPS (``length``): ``conv_post`` writes zeros beyond them, so
        the whole of ``o_hat`` is defined -- what lies between the two is computed from neighbours that were skipped
        and must not leak into the result."""
        B = lengths.shape[0]
        lim = torch.empty(2, B, dtype=torch.int32, device=self.device)
        _lib.call("ov_frame_limits_i32", lengths, lim[0], B, int(T), self.generator_margin)
        _lib.call("ov_frame_limits_i32", lengths, lim[1], B, int(T), 0)
        return lim

    def decode(self, z_hat, cond_d, ws=None, T=None, limits=None):
        """Generator (models.py:272-291).  ``z_hat`` is [B, C, ld] with ``T`` valid frames per row
        (``T`` defaults to the full row, i.e. a dense tensor).  ``limits`` (``frame_limits``): frames per utterance
        to compute -- time tiles wholly beyond are dropped from every launch's work list."""
        B, C, ld = z_hat.shape
        T = ld if T is None else T
        if ws is None:
            ws = self._workspace(B, T)
        Tp = ws["Tp"]
        cfg = self.cfg
        ch = cfg["upsample_initial_channel"]
        keep = None
        if limits is not None:
            limits, keep = limits[0], limits[1]
            if B > LIMIT_MAX_BATCH:
                # the conv kernels' prefix table holds 256 utterances: beyond that every launch computes the whole
                # tensors, but conv_post (a per-utterance limit, no table) still keeps `length` frames and writes zeros
                # beyond them -- the skip_padding contract ("the padded tail of o_hat is zero") holds at any batch size
                limits = None
        lim = lambda scale: dict(col_limit=limits, col_limit_scale=scale) if limits is not None else {}
        self._conv(self.conv_pre, z_hat, 0, C * ld, ws["pre"], 0, ch * Tp, B, T, bias_b=cond_d,
                   bias_b_bs=0 if cond_d.shape[0] == 1 else cond_d.shape[1], x_ld=ld, out_ld=Tp, tag="conv_pre",
                   **lim(1))
        x, L, x_ld = ws["pre"], T, Tp
        free = list(ws["dec"])
        nk = len(cfg["resblock_kernel_sizes"])
        rate = 1                                  # columns per frame at the current stage
        for i, up in enumerate(self.ups):
            s = up["stride"]
            cin, ch = ch, ch // 2
            u = free.pop()
            # leaky_relu(0.1) + ConvTranspose1d (models.py:278-279)
            self._conv(up["conv"], x, 0, cin * x_ld, u, 0, ch * L * s, B, L, epi=EPI_CONVT, in_slope=LRELU_SLOPE,
                       flags=up["flags"], phase_s=s, x_ld=x_ld, tag="ups", alg_flops=2.0 * cin * ch * 2 * s * L * B,
                       **lim(rate))
            rate *= s
            if i > 0:
                free.append(x)
            L *= s
            x_ld = L
            if self._split3_on and self.split_resblocks[i] is not None:
                # split-precision stage (opt-in); with limits its convs carry the same length-aware work lists
                acc = free.pop()
                self._mrf_split(self.split_resblocks[i], u, acc, ws, B, ch, L, limits=limits, rate=rate)
                free.append(u)
                x = acc
                continue
            t1, ra, acc = free.pop(), free.pop(), free.pop()
            bs = ch * L
            # MRF: mean of the 3 ResBlock1 outputs (models.py:280-286, modules.py:296-306).  The three ResBlocks of a
            # stage are independent chains until the sum.  At small batches one launch does not fill the chip (batch 1,
            # stage 0: 108 workgroups for 512 slots; stages 1-3: one partial round of 431), so there the chains run on
            # separate HIP streams and fill each other's ramps and tails; chain j's last launch waits for chain j - 1's
            # (the running sum keeps its order: bit-identical to the serial sequence).
            concurrent = (self.chain_streams > 1 and nk > 1 and B <= self.chain_streams_max_batch and self.profile is None
                          # the chains' extra scratch is allocated lazily: never from a capturing graph's private pool
                          and not (torch.cuda.is_current_stream_capturing() and len(ws.get("dec_extra", [])) < 2 * (nk - 1)))
            scratch = [(t1, ra)]
            if concurrent:
                extra = self._chain_scratch(ws, 2 * (nk - 1))
                scratch += [(extra[2 * j], extra[2 * j + 1]) for j in range(nk - 1)]

            def chain(j, pairs):
                t1_, ra_ = scratch[j if concurrent else 0]
                cur = u
                fused = self.fuse_pairs and L % 4 == 0 and all(
                    (ch, c1.K) in PAIR_POLICY and pair_supported(ch, c1.K, c1.dil) for c1, _ in pairs)
                # Winograd-domain convs where an instance exists and wino_policy() picks it; they carry the same
                # length-aware work lists (skip_padding) as the direct kernels
                wn = self.wino_resblocks[i][j] if (self.use_winograd and L % 4 == 0) else None
                for n, (c1, c2) in enumerate(pairs):
                    last = n == len(pairs) - 1
                    if last and concurrent and j > 0:
                        torch.cuda.current_stream(self.device).wait_event(done[j - 1])
                    add = acc if (last and j > 0) else None
                    scale = 1.0 / nk if (last and j == nk - 1) else 1.0
                    if fused:
                        # one launch per pair, intermediate in LDS; out must not alias x: ra / t1 ping-pong
                        dst = acc if last else (t1_ if cur is ra_ else ra_)
                        self._pair(c1, c2, cur, dst, bs, B, L, add, scale, **lim(rate))
                    else:
                        w1, w2 = wn[n] if wn is not None else (None, None)
                        if w1 is not None and wino_items(ch, w1.dil, B, L) < WINO_MIN_ITEMS:
                            w1 = None           # too few items for one workgroup per CU: the direct kernel's small tiles
                        if w2 is not None and wino_items(ch, 1, B, L) < WINO_MIN_ITEMS:
                            w2 = None
                        # t1 is read by c2 only, which activates it: a Winograd c1 stores it activated (the same
                        # values, modules.py:298-301) and c2 -- either kernel -- stages it as is
                        if w1 is not None:
                            self._wino(w1, cur, t1_, bs, B, L, out_slope=LRELU_SLOPE, **lim(rate))
                        else:
                            self._conv(c1, cur, 0, bs, t1_, 0, bs, B, L, in_slope=LRELU_SLOPE, tag="mrf", **lim(rate))
                        c2_slope = 1.0 if w1 is not None else LRELU_SLOPE
                        dst = acc if last else ra_
                        if w2 is not None:
                            self._wino(w2, t1_, dst, bs, B, L, res=cur, add=add, scale=scale, in_slope=c2_slope, **lim(rate))
                        else:
                            self._conv(c2, t1_, 0, bs, dst, 0, bs, B, L, in_slope=c2_slope, res=cur, res_bs=bs,
                                       add=add, add_bs=bs, scale=scale, tag="mrf", **lim(rate))
                    cur = dst

            if not concurrent:
                for j, pairs in enumerate(self.resblocks[i]):
                    chain(j, pairs)
            else:
                main = torch.cuda.current_stream(self.device)
                side = self._side_streams(nk)
                fork = torch.cuda.Event()
                fork.record(main)
                done = [None] * nk
                for j, pairs in enumerate(self.resblocks[i]):
                    with torch.cuda.stream(side[j]):
                        side[j].wait_event(fork)
                        chain(j, pairs)
                        done[j] = torch.cuda.Event()
                        done[j].record(side[j])
                for j in range(nk):                     # every chain's scratch is free again before the next stage
                    main.wait_event(done[j])
            free += [u, t1, ra]
            x = acc
        o_hat = torch.empty(B, 1, L, dtype=torch.float32, device=self.device)
        if keep is not None:
            _lib.call("ov_conv_post_tanh_limited_f32", x, self.post_w, o_hat, B, ch, L, self.post_w.shape[1],
                      FINAL_LRELU_SLOPE, keep, rate)
        else:
            _lib.call("ov_conv_post_tanh_f32", x, self.post_w, o_hat, B, ch, L, self.post_w.shape[1], FINAL_LRELU_SLOPE)
        return o_hat

    # ---- extract_se path -----------------------------------------------------------------------------
    @torch.no_grad()
    @on_own_device
    def reference_encoder(self, spec_t):
        """``spec_t`` [N, Ty, n_freq] (the transposed spectrogram the reference API passes,
        openvoice/api.py:131) -> [N, gin]; reference: openvoice/models.py:339-359.  Internally every
        tensor is [N][C][F][T] (time contiguous), i.e. the un-transposed spectrogram: when the caller
        passes ``spec.transpose(1, 2)`` the transpose back is a view and nothing is copied."""
        re = self.ref_enc
        if re is None:
            raise _lib.OvError("this checkpoint has no ref_enc.* weights")
        dev = self.device
        x = spec_t.to(dev, torch.float32).transpose(1, 2).contiguous()     # [N, F, T]
        N, F, T = x.shape
        assert F == self.spec_channels
        cur = torch.empty_like(x)
        _lib.call("ov_layernorm_freq_f32", x, re["ln_w"], re["ln_b"], cur, N, F, T, 1e-5)
        cin = 1
        for w, b in re["convs"]:
            cout = w.shape[0]
            Fo, To = (F - 1) // 2 + 1, (T - 1) // 2 + 1
            nxt = torch.empty(N, cout, Fo, To, dtype=torch.float32, device=dev)
            _lib.call("ov_conv2d_s2_relu_f32", cur, w, b, nxt, N, cin, cout, F, T)
            cur, cin, F, T = nxt, cout, Fo, To
        feat = cin * F                                                        # 128 * 9 = 1152
        H = REF_ENC_GRU
        gi = torch.empty(N, 3 * H, T, dtype=torch.float32, device=dev)
        self._conv(re["gru_in"], cur, 0, feat * T, gi, 0, 3 * H * T, N, T, tag="gru_in")
        h = torch.empty(N, H, dtype=torch.float32, device=dev)
        _lib.call("ov_gru_f32", gi, re["whh_t"], re["bhh"], h, N, H, T)
        return self._linear(h, re["proj_w"], re["proj_b"])


class GraphedConversion:
    """One conversion shape -- (B, T frames, tau, rows of src/tgt speaker embedding) -- captured once into a HIP graph
    and replayed.  The path is ~330 kernel launches with no host sync and no data-dependent control flow, so a
    replay is one ``hipGraphLaunch``: at batch 1 (what ``ToneColorConverter.convert`` issues per file) the eager
    path is bound by the ~10 us of Python/ctypes per launch, not by the GPU.

    Inputs are copied into static device buffers, outputs are static too: **they are overwritten by the next
    replay** -- clone what must outlive it (``ToneColorConverter.convert_batch`` does).  The engine's workspace of
    this shape is pinned for the life of the graph.  Reference contract unchanged: openvoice/models.py:492-499."""

    @on_own_device
    def __init__(self, engine, B, T, tau, src_rows=1, tgt_rows=1, skip_padding=False):
        dev = engine.device
        self.engine, self.B, self.T, self.tau = engine, int(B), int(T), float(tau)
        f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.spec = f(B, engine.spec_channels, padded_frames(T))[:, :, :T]      # 16-byte aligned rows
        self.lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
        self.g_src, self.g_tgt = f(src_rows, engine.gin, 1), f(tgt_rows, engine.gin, 1)
        self.noise = f(B, engine.inter, T)
        run = lambda: engine.voice_conversion(self.spec, self.lengths, self.g_src, self.g_tgt, tau=self.tau,
                                              noise=self.noise, skip_padding=skip_padding)
        saved, engine.profile = engine.profile, None       # event records are not capturable
        try:
            side = torch.cuda.Stream(dev)                      # torch's rule: warm up off the default stream
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                run()                                          # allocates the workspace, fills the occupancy caches
            torch.cuda.current_stream(dev).wait_stream(side)
            self.workspace = engine._workspace(B, T)           # keep it alive if the engine moves to another shape
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = run()
        finally:
            engine.profile = saved

    @torch.no_grad()
    @on_own_device
    def __call__(self, spec, spec_lengths, sid_src, sid_tgt, noise=None):
        if tuple(spec.shape) != (self.B, self.engine.spec_channels, self.T):
            raise _lib.OvError(f"graph captured for spec {(self.B, self.engine.spec_channels, self.T)}, got {tuple(spec.shape)}")
        self.spec.copy_(spec)
        self.lengths.copy_(spec_lengths)
        self.g_src.copy_(sid_src.reshape(self.g_src.shape))
        self.g_tgt.copy_(sid_tgt.reshape(self.g_tgt.shape))
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise)
        self.graph.replay()
        return self.out
