"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares, and
its host-side weight packer produces the documented MFMA fragment layout.  No kernel is launched."""
import os
import re

import numpy as np
import pytest
import torch

from openvoice_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH),
                                reason="libopenvoice_amd.so not built (run __graft_entry__.build())")


def test_every_header_symbol_is_exported_and_bound():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    declared = set(re.findall(r"\b(ov_[a-z0-9_]+)\s*\(", header))
    declared -= {"ov_conv1d_params"}
    lib = _lib.load()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.ov_version() >= 100


def test_conv_params_struct_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_conv1d_params {"):header.index("} ov_conv1d_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.ConvParams._fields_]


@pytest.mark.parametrize("cout,cin,k", [(32, 32, 3), (96, 192, 1), (384, 192, 5), (192, 513, 1), (64, 64, 11)])
def test_weight_packer_layout(cout, cin, k):
    """Record (mt, U, g), lane l, component u holds W[32mt + (l&31)][8U + 2p + (l>>5)][tap] with
    s = 4g + u, p = s // K, tap = s % K (DESIGN.md 'Packed weights')."""
    lib = _lib.load()
    gen = torch.Generator().manual_seed(cout * 1000 + cin + k)
    w = torch.randn(cout, cin, k, generator=gen)
    n = lib.ov_conv1d_pack_size(cout, cin, k)
    dst = torch.full((n,), float("nan"))
    assert lib.ov_conv1d_pack_f32(w.data_ptr(), cout, cin, k, dst.data_ptr()) == 0
    rows = lib.ov_conv1d_pack_rows(cout)
    assert rows % 128 == 0 and rows >= cout
    nu = ((cin + 7) // 8 + 3) // 4 * 4       # units padded to the largest units-per-chunk
    recs = nu * k + 1
    assert n == rows // 32 * recs * 256
    got = dst.numpy().reshape(rows // 32, recs, 64, 4)
    assert not np.isnan(got).any()
    wp = np.zeros((rows, nu * 8, k), dtype=np.float32)
    wp[:cout, :cin] = w.numpy()
    lane = np.arange(64)
    for mt in range(rows // 32):
        for U in range(nu):
            for g in range(k):
                for u in range(4):
                    s = 4 * g + u
                    p, tap = divmod(s, k)
                    exp = wp[32 * mt + (lane & 31), 8 * U + 2 * p + (lane >> 5), tap]
                    assert np.array_equal(got[mt, U * k + g, :, u], exp), (mt, U, g, u)
    assert np.all(got[:, -1] == 0)    # prefetch-overrun record


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.load()
    assert lib.ov_conv1d_f32(None, None) == -1
    p = _lib.ConvParams()
    assert lib.ov_conv1d_f32(p, None) == -1
    assert lib.ov_linear_f32(None, None, None, None, 1, 1, 1, None) == -1
    assert lib.ov_conv_post_tanh_f32(None, None, None, 1, 1, 1, 7, 0.01, None) == -1
    assert lib.ov_sequence_mask_f32(None, None, 1, 1, 0, None) == -1
    assert lib.ov_conv1d_pack_f32(None, 1, 1, 1, None) == -1
    assert lib.ov_conv1d_pack_size(0, 1, 1) == 0


def test_bf16_params_struct_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_conv1d_bf16_params {"):header.index("} ov_conv1d_bf16_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.ConvBf16Params._fields_]


@pytest.mark.parametrize("cout,cin,k", [(32, 32, 3), (64, 96, 7), (96, 64, 1), (256, 32, 11)])
def test_bf16_weight_packer_layout_and_rounding(cout, cin, k):
    """Record ((nt * chunks + c) * K + tap) * 2 + kb, lane l, element i holds bf16(W[32nt + (l&31)][32c + 16kb +
    8(l>>5) + i][tap]), rounded to nearest even exactly like torch.bfloat16."""
    lib = _lib.load()
    gen = torch.Generator().manual_seed(cout + cin + k)
    w = torch.randn(cout, cin, k, generator=gen)
    n = lib.ov_conv1d_bf16_pack_size(cout, cin, k)
    ntiles, chunks = (cout + 31) // 32, cin // 32
    assert n == (ntiles * chunks * k * 2 + 1) * 512
    dst = torch.full((n,), -1, dtype=torch.int16)
    assert lib.ov_conv1d_bf16_pack(w.data_ptr(), cout, cin, k, dst.data_ptr()) == 0
    got = dst.view(torch.bfloat16).float().reshape(ntiles * chunks * k * 2 + 1, 64, 8)
    want = torch.zeros(ntiles * 32, cin, k)
    want[:cout] = w.to(torch.bfloat16).float()
    lane = torch.arange(64)
    for nt in range(ntiles):
        for c in range(chunks):
            for tap in range(k):
                for kb in range(2):
                    rec = got[((nt * chunks + c) * k + tap) * 2 + kb]
                    for i in range(8):
                        exp = want[32 * nt + (lane & 31), 32 * c + 16 * kb + 8 * (lane >> 5) + i, tap]
                        assert torch.equal(rec[:, i], exp), (nt, c, tap, kb, i)
    assert torch.all(got[-1] == 0)                         # the zero record that ends the stream
    assert lib.ov_conv1d_bf16_pack_size(32, 40, 3) == 0     # Cin must be a multiple of 32
    assert lib.ov_conv1d_bf16cl(None, None) == -1


@pytest.mark.parametrize("c,k", [(32, 3), (64, 7), (128, 11)])
def test_bf16_pair2_weight_stream_is_in_16x16x32_fragment_order(c, k):
    """ov_conv1d_bf16_pack16 (the fused pair's weights): record ((nt * chunks + ch) * K + tap) * 2 + f, lane l, element i
    holds bf16(W[32 nt + 16 f + (l & 15)][32 ch + 8 (l >> 4) + i][tap]) -- the A operand of v_mfma_f32_16x16x32_bf16 --
    the same multiset of values as ov_conv1d_bf16_pack's stream, one trailing zero record."""
    lib = _lib.load()
    w = torch.randn(c, c, k, generator=torch.Generator().manual_seed(c + k))
    n = lib.ov_conv1d_bf16_pack_size(c, c, k)
    dst, old = torch.full((n,), -1, dtype=torch.int16), torch.full((n,), -1, dtype=torch.int16)
    assert lib.ov_conv1d_bf16_pack16(w.data_ptr(), c, c, k, dst.data_ptr()) == 0
    assert lib.ov_conv1d_bf16_pack(w.data_ptr(), c, c, k, old.data_ptr()) == 0
    ntiles = chunks = c // 32
    got = dst.view(torch.bfloat16).float().reshape(ntiles * chunks * k * 2 + 1, 64, 8)
    want = w.to(torch.bfloat16).float()
    lane = torch.arange(64)
    for nt in range(ntiles):
        for ch in range(chunks):
            for tap in range(k):
                for f in range(2):
                    rec = got[((nt * chunks + ch) * k + tap) * 2 + f]
                    for i in range(8):
                        exp = want[32 * nt + 16 * f + (lane & 15), 32 * ch + 8 * (lane >> 4) + i, tap]
                        assert torch.equal(rec[:, i], exp), (nt, ch, tap, f, i)
    assert torch.all(got[-1] == 0)
    assert torch.equal(dst.sort().values, old.sort().values)
    assert lib.ov_conv1d_bf16_pack16(None, c, c, k, dst.data_ptr()) == -1
    assert lib.ov_conv1d_bf16_pack16(w.data_ptr(), c, 40, k, dst.data_ptr()) == -1   # Cin must be a multiple of 32


def test_respair_params_struct_matches_header_field_order_and_size():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_respair_params {"):header.index("} ov_respair_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.RespairParams._fields_]
    import ctypes
    assert ctypes.sizeof(_lib.RespairParams) == 144          # 7 pointers, 3 int64, 7 int32, 2 float, pad, 2 pointers, 2 int32
    lib = _lib.load()
    assert lib.ov_resblock_pair_f32(None, None) == -1
    assert lib.ov_resblock_pair_supported(32, 3, 1) == 1 and lib.ov_resblock_pair_supported(128, 3, 1) == 0


def test_respair_bf16_params_struct_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_respair_bf16_params {"):header.index("} ov_respair_bf16_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.RespairBf16Params._fields_]
    lib = _lib.load()
    assert lib.ov_resblock_pair_bf16cl(None, None) == -1
    assert lib.ov_resblock_pair_bf16_supported(32, 11, 5) == 1 and lib.ov_resblock_pair_bf16_supported(64, 7, 1) == 0


def test_respair2_bf16_params_struct_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_respair2_bf16_params {"):header.index("} ov_respair2_bf16_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.Respair2Bf16Params._fields_]
    import ctypes
    assert ctypes.sizeof(_lib.Respair2Bf16Params) == 104      # 7 pointers, 6 int32, 3 float, 1 int32, 1 pointer
    lib = _lib.load()
    assert lib.ov_resblock_pair2_bf16cl(None, None) == -1
    assert lib.ov_resblock_pair2_bf16_supported(128, 11, 5) == 1 and lib.ov_resblock_pair2_bf16_supported(64, 3, 1) == 1
    assert lib.ov_resblock_pair2_bf16_supported(32, 3, 1) == 1 and lib.ov_resblock_pair2_bf16_supported(256, 7, 1) == 0


def test_every_header_function_is_bound_and_exported():
    """Every `int|size_t ov_*(...)` declared in include/openvoice_amd.h is in _lib.SIGNATURES and exported by the .so."""
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"^(?:int|size_t)\s+(ov_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)


def test_production_library_is_not_a_measurement_build():
    assert _lib.load().ov_build_experiment() == 0


def test_torch_binding_loads_without_a_gpu_and_rejects_cpu_tensors():
    """openvoice_amd/libopenvoice_amd_torch.so (csrc/torch_shim.cpp): TORCH_LIBRARY ops over the same C ABI."""
    import torch
    ops = _lib.torch_ops()
    assert ops.version() == _lib.load().ov_version()
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.linear_f32(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3), torch.zeros(2, 3), 2, 3, 4)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.sequence_mask_f32(torch.zeros(2, dtype=torch.long), torch.zeros(2, 8), 2, 8, 8)


def test_torch_binding_has_one_op_per_header_function():
    """`torch.ops.openvoice_amd.<name>` exists for every `ov_<name>` of include/openvoice_amd.h (the shim derives the
    flat-argument ops from the C prototypes, so the argument lists cannot drift), and the schema of a derived op maps
    pointers to optional tensors -- mutable ones annotated -- and scalars to int / float."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "openvoice_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"^(?:int|size_t)\s+(ov_\w+)\s*\(", header, flags=re.M))
    ops = _lib.torch_ops()
    for name in declared:
        assert hasattr(ops, name[3:]), f"torch.ops.openvoice_amd.{name[3:]} missing"
    schema = str(ops.gru_f32.default._schema)
    assert schema == ("openvoice_amd::gru_f32(Tensor? a0, Tensor? a1, Tensor? a2, Tensor(a3!)? a3, int a4, int a5, "
                      "int a6) -> ()"), schema
    assert str(ops.wn_layer_tile.default._schema).endswith("-> int")


def test_host_functions_agree_between_the_two_bindings(monkeypatch):
    """Weight packers and capability queries through `_lib.call` under both bindings (host functions: no GPU needed)."""
    import torch
    w = torch.randn(64, 32, 3, generator=torch.Generator().manual_seed(0))
    got = {}
    for binding in ("ctypes", "torch"):
        monkeypatch.setenv("OPENVOICE_AMD_BINDING", binding)
        n = _lib.call("ov_conv1d_pack_size", 64, 32, 3)
        dst = torch.zeros(n)
        _lib.call("ov_conv1d_pack_f32", w, 64, 32, 3, dst)
        nb = _lib.call("ov_conv1d_bf16_pack_size", 64, 32, 3)
        dstb = torch.zeros(nb, dtype=torch.int16)
        _lib.call("ov_conv1d_bf16_pack", w, 64, 32, 3, dstb)
        got[binding] = (n, dst, nb, dstb, _lib.call("ov_resblock_pair_supported", 32, 3, 1),
                        _lib.call("ov_wn_layer_supported", 192, 5), _lib.call("ov_conv1d_pack_rows", 64))
        with pytest.raises(_lib.OvError):
            _lib.call("ov_conv1d_pack_f32", None, 64, 32, 3, dst)
    a, b = got["ctypes"], got["torch"]
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and a[2] == b[2] and torch.equal(a[3], b[3]) and a[4:] == b[4:]
    monkeypatch.setenv("OPENVOICE_AMD_BINDING", "pybind")
    with pytest.raises(_lib.OvError, match="expected 'torch' or 'ctypes'"):
        _lib.binding()


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/openvoice_amd.h is the contract a non-Python host binds: it must compile as C (gcc -std=c99, what cgo /
    JNI / any FFI generator feeds on) and as C++, with no torch or HIP headers in sight."""
    import shutil
    import subprocess
    header = os.path.join(REPO, "include", "openvoice_amd.h")
    src = tmp_path / "use.c"
    src.write_text('#include "openvoice_amd.h"\n'
                   'int probe(void) { ov_conv1d_params p; ov_respair_params q; ov_wn_layer_params w;\n'
                   '  p.col_limit = 0; q.col_limit = 0; w.dbg = 0; (void)p; (void)q; (void)w;\n'
                   '  return (int)sizeof(ov_conv1d_params) + OV_E_ALIGN + OV_EPI_MAGNITUDE; }\n')
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        r = subprocess.run([cc, std, "-x", lang, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                            "-I", os.path.dirname(header), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_conv_split3_params_struct_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_conv1d_split3_params {"):header.index("} ov_conv1d_split3_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.ConvSplit3Params._fields_]
    import ctypes
    assert ctypes.sizeof(_lib.ConvSplit3Params) == 128     # 5 pointers, 3 int64, 8 int32, 3 float, 1 int32, 2 pointers
    lib = _lib.load()
    assert lib.ov_conv1d_split3(None, None) == -1
    assert lib.ov_version() >= _lib.MIN_VERSION >= 206


def test_conv_wino_params_struct_matches_header_field_order():
    """ABI 2.07: ``ov_conv1d_wino_params`` field for field against the ctypes mirror, and the entry point's argument checks
    that need no GPU."""
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    body = header[header.index("typedef struct ov_conv1d_wino_params {"):header.index("} ov_conv1d_wino_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)$", first.strip())[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in _lib.ConvWinoParams._fields_]
    import ctypes
    assert ctypes.sizeof(_lib.ConvWinoParams) == 6 * 8 + 4 * 8 + 10 * 4 + 2 * 4 + 8 + 8 + 2 * 4
    lib = _lib.load()
    assert lib.ov_conv1d_wino_f32(None, None) == -1
    assert lib.ov_version() >= 209
    assert [lib.ov_conv1d_wino_chunk(k, 128) for k in (3, 5, 7, 11)] == [16, 0, 8, 8]
    assert [lib.ov_conv1d_wino_chunk(k, 64) for k in (3, 7, 11)] == [8, 4, 4]
    # one 32-row fragment per workgroup: K = 11 only (a two-channel chunk of K = 7 would be an odd number of k-steps)
    assert [lib.ov_conv1d_wino_chunk(k, 32) for k in (3, 7, 11)] == [0, 0, 2] and lib.ov_conv1d_wino_chunk(11, 96) == 2
    assert lib.ov_conv1d_wino_chunk(11, 48) == 0


def test_header_abi_version_macro_matches_the_library():
    header = open(os.path.join(REPO, "include", "openvoice_amd.h")).read()
    macro = int(re.search(r"#define OV_ABI_VERSION (\d+)", header).group(1))
    assert _lib.load().ov_version() == macro == _lib.MIN_VERSION
