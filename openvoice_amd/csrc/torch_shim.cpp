// Thin torch binding of the C ABI ("thin pybind/TORCH_LIBRARY shim converts at::Tensor";
// BASELINE.json north_star: "bound through a thin torch cpp_extension C-ABI").  `torch.ops.openvoice_amd.<name>` exists
// for EVERY entry point `ov_<name>` of include/openvoice_amd.h -- tests/test_abi_cpu.py checks the two lists against each
// other -- takes tensors where the C function takes pointers, checks device / dtype with TORCH_CHECK, picks up
// c10::hip::getCurrentHIPStream() of the tensors' device and calls the extern "C" function; a non-zero OV_E_* code
// becomes a RuntimeError.  The library itself stays free of torch types -- this file is the only one that sees both
// sides.
//
// Two kinds of wrapper:
//  * entry points with a flat argument list are bound by ONE variadic adapter (`Bind<&ov_fn>`) that derives the torch
//    signature and the schema string from the C prototype itself (pointer -> `Tensor?`, `T*` non-const ->
//    `Tensor(a!)?`, int / int64_t -> `int`, float -> `float`, trailing ov_stream_t -> the current stream), so the shim
//    cannot drift from the header: a changed prototype changes the op, a mismatch does not compile;
//  * the entry points that take a parameter struct (`ov_conv1d_f32`, `ov_resblock_pair_f32`, `ov_wn_layer_f32`,
//    `ov_conv1d_bf16cl`, `ov_resblock_pair_bf16cl`, `ov_resblock_pair2_bf16cl`, `ov_conv1d_split3`) take the struct's pointers as tensors and its integers / floats as
//    `int[]` / `float[]` in declaration order.
// Pointer + element offset (a channel or row-block offset into a larger tensor) is expressed by the caller as a view.
//
// Built by `make -C openvoice_amd/csrc torch_shim` (hipcc, in-tree: openvoice_amd/libopenvoice_amd_torch.so).  This is
// the DEFAULT binding of the Python package (openvoice_amd/_lib.py); OPENVOICE_AMD_BINDING=ctypes selects the
// libtorch-free ctypes binding of the same C ABI.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>
#include <type_traits>

#include "openvoice_amd.h"

namespace {

const char* ov_strerror(int rc) {
  switch (rc) {
    case OV_E_BADARG: return "OV_E_BADARG";
    case OV_E_UNSUPPORTED: return "OV_E_UNSUPPORTED";
    case OV_E_ALIGN: return "OV_E_ALIGN";
    case OV_E_LAUNCH: return "OV_E_LAUNCH";
    default: return "unknown error";
  }
}

using OptTensor = c10::optional<at::Tensor>;

// Per-call context: the device every tensor argument must live on (first tensor seen decides) and whether the entry
// point is a host function (no stream parameter: CPU tensors) or a device function.
struct Ctx {
  const char* name;
  bool host;
  bool have_dev = false;
  c10::Device dev{c10::kCPU};
  void see(const at::Tensor& t, int index) {
    if (host) {
      TORCH_CHECK(t.device().is_cpu(), name, ": argument ", index, " must be a HOST (CPU) tensor");
      return;
    }
    TORCH_CHECK(t.is_cuda(), name, ": argument ", index, " must be on a ROCm device (there is no CPU path)");
    if (!have_dev) { dev = t.device(); have_dev = true; }
    TORCH_CHECK(t.device() == dev, name, ": argument ", index, " is on ", t.device(), ", expected ", dev);
  }
  ov_stream_t stream() const {
    TORCH_CHECK(have_dev, name, ": no device tensor among the arguments");
    return static_cast<ov_stream_t>(c10::hip::getCurrentHIPStream(dev.index()).stream());
  }
};

// The launch happens with the tensors' device current (the library sizes launches for, and launches on, the CURRENT
// device): a caller holding tensors on cuda:1 while cuda:0 is current gets the same behaviour as any torch op.
struct DeviceScope {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;   // ROCm torch tensors report DeviceType "cuda"
  explicit DeviceScope(const Ctx& c) { if (c.have_dev) guard.set_device(c.dev); }
};

template <typename E> bool dtype_ok(at::ScalarType s);
template <> bool dtype_ok<float>(at::ScalarType s) { return s == at::kFloat; }
template <> bool dtype_ok<double>(at::ScalarType s) { return s == at::kDouble; }
template <> bool dtype_ok<int64_t>(at::ScalarType s) { return s == at::kLong; }
template <> bool dtype_ok<int32_t>(at::ScalarType s) { return s == at::kInt; }
template <> bool dtype_ok<uint16_t>(at::ScalarType s) { return s == at::kBFloat16 || s == at::kShort || s == at::kUInt16; }
template <> bool dtype_ok<unsigned long long>(at::ScalarType s) { return s == at::kLong || s == at::kUInt64; }
template <typename E> const char* dtype_name();
template <> const char* dtype_name<float>() { return "float32"; }
template <> const char* dtype_name<double>() { return "float64"; }
template <> const char* dtype_name<int64_t>() { return "int64"; }
template <> const char* dtype_name<int32_t>() { return "int32"; }
template <> const char* dtype_name<uint16_t>() { return "bfloat16 (or int16 bit patterns)"; }
template <> const char* dtype_name<unsigned long long>() { return "int64 (tick counters)"; }

template <typename E> E* tensor_ptr(const OptTensor& t, Ctx& c, int index) {
  if (!t.has_value() || !t->defined()) return nullptr;        // the C function rejects a required NULL itself
  c.see(*t, index);
  TORCH_CHECK(dtype_ok<E>(t->scalar_type()), c.name, ": argument ", index, " must be ", dtype_name<E>(), ", got ",
              t->scalar_type());
  return static_cast<E*>(t->data_ptr());
}

// C parameter type -> torch parameter type, schema fragment, conversion
template <typename C, typename = void> struct Arg;
template <typename E> struct Arg<const E*> {
  using torch_t = const OptTensor&;
  static std::string schema(int i) { return "Tensor? a" + std::to_string(i); }
  static const E* get(const OptTensor& t, Ctx& c, int i) { return tensor_ptr<E>(t, c, i); }
};
template <typename E> struct Arg<E*, std::enable_if_t<!std::is_const<E>::value>> {
  using torch_t = const OptTensor&;
  static std::string schema(int i) { return "Tensor(a" + std::to_string(i) + "!)? a" + std::to_string(i); }
  static E* get(const OptTensor& t, Ctx& c, int i) { return tensor_ptr<E>(t, c, i); }
};
template <> struct Arg<int> {
  using torch_t = int64_t;
  static std::string schema(int i) { return "int a" + std::to_string(i); }
  static int get(int64_t v, Ctx& c, int i) {
    TORCH_CHECK(v >= INT32_MIN && v <= INT32_MAX, c.name, ": argument ", i, " does not fit a 32-bit int");
    return static_cast<int>(v);
  }
};
template <> struct Arg<int64_t> {
  using torch_t = int64_t;
  static std::string schema(int i) { return "int a" + std::to_string(i); }
  static int64_t get(int64_t v, Ctx&, int) { return v; }
};
template <> struct Arg<float> {
  using torch_t = double;
  static std::string schema(int i) { return "float a" + std::to_string(i); }
  static float get(double v, Ctx&, int) { return static_cast<float>(v); }
};

template <typename... T> struct List {};
template <typename L, typename... Acc> struct DropLast;
template <typename T, typename... Acc> struct DropLast<List<T>, Acc...> { using type = List<Acc...>; using last = T; };
template <typename T, typename U, typename... R, typename... Acc>
struct DropLast<List<T, U, R...>, Acc...> : DropLast<List<U, R...>, Acc..., T> {};

template <typename... C> std::string schema_of(const char* name, bool returns_int) {
  std::string s = std::string(name) + "(";
  int i = 0;
  bool first = true;
  ((s += (first ? "" : ", ") + Arg<C>::schema(i++), first = false), ...);
  return s + (returns_int ? ") -> int" : ") -> ()");
}

// kind: 0 = device function returning a status (last C parameter is the stream), 1 = host function returning a
// status, 2 = host function returning a value (sizes, capability queries, version)
template <int Kind, typename R, typename L> struct Wrap;
template <typename R, typename... C> struct Wrap<0, R, List<C...>> {
  template <R (*Fn)(C..., ov_stream_t)> struct On {
    static inline const char* name = "";
    static void call(typename Arg<C>::torch_t... a) {
      Ctx c{name, false};
      int i = 0;
      // braced init list: arguments are converted left to right
      std::tuple<C...> v{Arg<C>::get(a, c, i++)...};
      DeviceScope scope(c);
      const int rc = std::apply([&](C... x) { return Fn(x..., c.stream()); }, v);
      TORCH_CHECK(rc == OV_OK, "ov_", name, " failed: ", ov_strerror(rc));
    }
    static void def(torch::Library& m, const char* n) { name = n; m.def(schema_of<C...>(n, false).c_str(), &call); }
  };
};
template <typename R, typename... C> struct Wrap<1, R, List<C...>> {
  template <R (*Fn)(C...)> struct On {
    static inline const char* name = "";
    static void call(typename Arg<C>::torch_t... a) {
      Ctx c{name, true};
      int i = 0;
      std::tuple<C...> v{Arg<C>::get(a, c, i++)...};
      const int rc = std::apply(Fn, v);
      TORCH_CHECK(rc == OV_OK, "ov_", name, " failed: ", ov_strerror(rc));
    }
    static void def(torch::Library& m, const char* n) { name = n; m.def(schema_of<C...>(n, false).c_str(), &call); }
  };
};
template <typename R, typename... C> struct Wrap<2, R, List<C...>> {
  template <R (*Fn)(C...)> struct On {
    static inline const char* name = "";
    static int64_t call(typename Arg<C>::torch_t... a) {
      Ctx c{name, true};
      int i = 0;
      std::tuple<C...> v{Arg<C>::get(a, c, i++)...};
      return static_cast<int64_t>(std::apply(Fn, v));
    }
    static void def(torch::Library& m, const char* n) { name = n; m.def(schema_of<C...>(n, true).c_str(), &call); }
  };
};

template <typename F> struct Sig;
template <typename R, typename... C> struct Sig<R (*)(C...)> { using ret = R; using args = List<C...>; };

template <auto Fn> void bind_device(torch::Library& m, const char* name) {
  using S = Sig<decltype(Fn)>;
  using D = DropLast<typename S::args>;
  static_assert(std::is_same<typename D::last, ov_stream_t>::value, "device entry points end with the stream");
  Wrap<0, typename S::ret, typename D::type>::template On<Fn>::def(m, name);
}
template <auto Fn> void bind_host(torch::Library& m, const char* name) {
  using S = Sig<decltype(Fn)>;
  Wrap<1, typename S::ret, typename S::args>::template On<Fn>::def(m, name);
}
template <auto Fn> void bind_value(torch::Library& m, const char* name) {
  using S = Sig<decltype(Fn)>;
  Wrap<2, typename S::ret, typename S::args>::template On<Fn>::def(m, name);
}

// ---- the parameter-struct entry points ----------------------------------------------------------------------------
template <typename E> E* sptr(const OptTensor& t, Ctx& c, int index, int64_t off = 0) {
  E* p = tensor_ptr<E>(t, c, index);
  return p ? p + off : nullptr;
}
void finish(int rc, const char* what) { TORCH_CHECK(rc == OV_OK, what, " failed: ", ov_strerror(rc)); }

// ip = [B, Cin, L, x_ld, out_ld, M, Cout, K, dil, epi, flags, split, phase_s, tiles_per_wg, tile, loaders, chunk,
//       x_bstride, out_bstride, res_bstride, add_bstride, out2_bstride, bias_b_bstride, mask_bstride,
//       x_off, out_off, res_off, bias_b_off, col_limit_scale]   (strides and offsets in elements);
// fp = [in_slope, scale]
void conv1d_f32(const OptTensor& x, const OptTensor& w, const OptTensor& bias, const OptTensor& out, const OptTensor& res,
                const OptTensor& add, const OptTensor& out2, const OptTensor& mask, const OptTensor& bias_b,
                const OptTensor& col_limit, at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 29 && fp.size() == 2, "openvoice_amd::conv1d_f32: 29 integer and 2 float parameters");
  Ctx c{"conv1d_f32", false};
  ov_conv1d_params p{};
  p.x = sptr<float>(x, c, 0, ip[24]);
  p.w = sptr<float>(w, c, 1);
  p.bias = sptr<float>(bias, c, 2);
  p.out = sptr<float>(out, c, 3, ip[25]);
  p.res = sptr<float>(res, c, 4, ip[26]);
  p.add = sptr<float>(add, c, 5);
  p.out2 = sptr<float>(out2, c, 6);
  p.mask = sptr<float>(mask, c, 7);
  p.bias_b = sptr<float>(bias_b, c, 8, ip[27]);
  p.col_limit = sptr<int32_t>(col_limit, c, 9);
  p.col_limit_scale = (int32_t)ip[28];
  p.B = (int32_t)ip[0]; p.Cin = (int32_t)ip[1]; p.L = (int32_t)ip[2]; p.x_ld = (int32_t)ip[3]; p.out_ld = (int32_t)ip[4];
  p.M = (int32_t)ip[5]; p.Cout = (int32_t)ip[6]; p.K = (int32_t)ip[7]; p.dil = (int32_t)ip[8]; p.epi = (int32_t)ip[9];
  p.flags = (int32_t)ip[10]; p.split = (int32_t)ip[11]; p.phase_s = (int32_t)ip[12]; p.tiles_per_wg = (int32_t)ip[13];
  p.tile = (int32_t)ip[14]; p.loaders = (int32_t)ip[15]; p.chunk = (int32_t)ip[16];
  p.x_bstride = ip[17]; p.out_bstride = ip[18]; p.res_bstride = ip[19]; p.add_bstride = ip[20];
  p.out2_bstride = ip[21]; p.bias_b_bstride = ip[22]; p.mask_bstride = ip[23];
  p.in_slope = (float)fp[0]; p.scale = (float)fp[1];
  DeviceScope scope(c);
  finish(ov_conv1d_f32(&p, c.stream()), "ov_conv1d_f32");
}

// ip = [B, C, L, ld, K, dil, nwg, x_bstride, out_bstride, add_bstride, col_limit_scale];  fp = [slope, scale]
void resblock_pair_f32(const OptTensor& x, const OptTensor& w1, const OptTensor& b1, const OptTensor& w2,
                       const OptTensor& b2, const OptTensor& out, const OptTensor& add, const OptTensor& dbg,
                       const OptTensor& col_limit, at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 11 && fp.size() == 2, "openvoice_amd::resblock_pair_f32: 11 integer and 2 float parameters");
  Ctx c{"resblock_pair_f32", false};
  ov_respair_params p{};
  p.x = sptr<float>(x, c, 0); p.w1 = sptr<float>(w1, c, 1); p.b1 = sptr<float>(b1, c, 2);
  p.w2 = sptr<float>(w2, c, 3); p.b2 = sptr<float>(b2, c, 4); p.out = sptr<float>(out, c, 5);
  p.add = sptr<float>(add, c, 6);
  p.dbg = sptr<unsigned long long>(dbg, c, 7);
  p.col_limit = sptr<int32_t>(col_limit, c, 8);
  p.B = (int32_t)ip[0]; p.C = (int32_t)ip[1]; p.L = (int32_t)ip[2]; p.ld = (int32_t)ip[3]; p.K = (int32_t)ip[4];
  p.dil = (int32_t)ip[5]; p.nwg = (int32_t)ip[6];
  p.x_bstride = ip[7]; p.out_bstride = ip[8]; p.add_bstride = ip[9]; p.col_limit_scale = (int32_t)ip[10];
  p.slope = (float)fp[0]; p.scale = (float)fp[1];
  DeviceScope scope(c);
  finish(ov_resblock_pair_f32(&p, c.stream()), "ov_resblock_pair_f32");
}

// ip = [B, H, T, ld, K, first, last, width, bstride, cond_bstride, mask_bstride, cond_off, row_split]
void wn_layer_f32(const OptTensor& x, const OptTensor& out, const OptTensor& skip, const OptTensor& w_in,
                  const OptTensor& b_in, const OptTensor& cond, const OptTensor& w_rs, const OptTensor& b_rs,
                  const OptTensor& mask, const OptTensor& dbg, const OptTensor& acts, at::IntArrayRef ip) {
  TORCH_CHECK(ip.size() == 13, "openvoice_amd::wn_layer_f32: 13 integer parameters");
  Ctx c{"wn_layer_f32", false};
  ov_wn_layer_params p{};
  p.x = sptr<float>(x, c, 0); p.out = sptr<float>(out, c, 1); p.skip = sptr<float>(skip, c, 2);
  p.w_in = sptr<float>(w_in, c, 3); p.b_in = sptr<float>(b_in, c, 4); p.cond = sptr<float>(cond, c, 5, ip[11]);
  p.w_rs = sptr<float>(w_rs, c, 6); p.b_rs = sptr<float>(b_rs, c, 7); p.mask = sptr<float>(mask, c, 8);
  p.dbg = sptr<unsigned long long>(dbg, c, 9);
  p.acts = sptr<float>(acts, c, 10);
  p.B = (int32_t)ip[0]; p.H = (int32_t)ip[1]; p.T = (int32_t)ip[2]; p.ld = (int32_t)ip[3]; p.K = (int32_t)ip[4];
  p.first = (int32_t)ip[5]; p.last = (int32_t)ip[6]; p.width = (int32_t)ip[7];
  p.bstride = ip[8]; p.cond_bstride = ip[9]; p.mask_bstride = ip[10]; p.row_split = (int32_t)ip[12];
  DeviceScope scope(c);
  finish(ov_wn_layer_f32(&p, c.stream()), "ov_wn_layer_f32");
}

// ip = [B, L, Cin, Cout, K, dil, phase_s, bias_bstride, layout];  fp = [in_slope, scale, out_slope]
void conv1d_bf16cl(const OptTensor& x, const OptTensor& w, const OptTensor& bias, const OptTensor& out,
                   const OptTensor& res, const OptTensor& add, const OptTensor& dbg, at::IntArrayRef ip,
                   at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 9 && fp.size() == 3, "openvoice_amd::conv1d_bf16cl: 9 integer and 3 float parameters");
  Ctx c{"conv1d_bf16cl", false};
  ov_conv1d_bf16_params p{};
  p.x = sptr<uint16_t>(x, c, 0); p.w = sptr<uint16_t>(w, c, 1); p.bias = sptr<float>(bias, c, 2);
  p.out = sptr<uint16_t>(out, c, 3); p.res = sptr<uint16_t>(res, c, 4); p.add = sptr<uint16_t>(add, c, 5);
  p.dbg = sptr<unsigned long long>(dbg, c, 6);
  p.B = (int32_t)ip[0]; p.L = (int32_t)ip[1]; p.Cin = (int32_t)ip[2]; p.Cout = (int32_t)ip[3]; p.K = (int32_t)ip[4];
  p.dil = (int32_t)ip[5]; p.phase_s = (int32_t)ip[6]; p.bias_bstride = (int32_t)ip[7]; p.layout = (int32_t)ip[8];
  p.in_slope = (float)fp[0]; p.scale = (float)fp[1]; p.out_slope = (float)fp[2];
  DeviceScope scope(c);
  finish(ov_conv1d_bf16cl(&p, c.stream()), "ov_conv1d_bf16cl");
}

// ip = [B, L, C, K, dil, nwg];  fp = [slope, scale]
void resblock_pair_bf16cl(const OptTensor& x, const OptTensor& w1, const OptTensor& b1, const OptTensor& w2,
                          const OptTensor& b2, const OptTensor& out, const OptTensor& add, const OptTensor& dbg,
                          at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 6 && fp.size() == 2, "openvoice_amd::resblock_pair_bf16cl: 6 integer and 2 float parameters");
  Ctx c{"resblock_pair_bf16cl", false};
  ov_respair_bf16_params p{};
  p.x = sptr<uint16_t>(x, c, 0); p.w1 = sptr<uint16_t>(w1, c, 1); p.b1 = sptr<float>(b1, c, 2);
  p.w2 = sptr<uint16_t>(w2, c, 3); p.b2 = sptr<float>(b2, c, 4); p.out = sptr<uint16_t>(out, c, 5);
  p.add = sptr<uint16_t>(add, c, 6);
  p.dbg = sptr<unsigned long long>(dbg, c, 7);
  p.B = (int32_t)ip[0]; p.L = (int32_t)ip[1]; p.C = (int32_t)ip[2]; p.K = (int32_t)ip[3]; p.dil = (int32_t)ip[4];
  p.nwg = (int32_t)ip[5];
  p.slope = (float)fp[0]; p.scale = (float)fp[1];
  DeviceScope scope(c);
  finish(ov_resblock_pair_bf16cl(&p, c.stream()), "ov_resblock_pair_bf16cl");
}

// ip = [B, L, C, K, dil, nwg, exp_flags];  fp = [slope, scale, out_slope]
void resblock_pair2_bf16cl(const OptTensor& x, const OptTensor& w1, const OptTensor& b1, const OptTensor& w2,
                           const OptTensor& b2, const OptTensor& out, const OptTensor& add, const OptTensor& dbg,
                           at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 7 && fp.size() == 3, "openvoice_amd::resblock_pair2_bf16cl: 7 integer and 3 float parameters");
  Ctx c{"resblock_pair2_bf16cl", false};
  ov_respair2_bf16_params p{};
  p.x = sptr<uint16_t>(x, c, 0); p.w1 = sptr<uint16_t>(w1, c, 1); p.b1 = sptr<float>(b1, c, 2);
  p.w2 = sptr<uint16_t>(w2, c, 3); p.b2 = sptr<float>(b2, c, 4); p.out = sptr<uint16_t>(out, c, 5);
  p.add = sptr<uint16_t>(add, c, 6);
  p.dbg = sptr<unsigned long long>(dbg, c, 7);
  p.B = (int32_t)ip[0]; p.L = (int32_t)ip[1]; p.C = (int32_t)ip[2]; p.K = (int32_t)ip[3]; p.dil = (int32_t)ip[4];
  p.nwg = (int32_t)ip[5]; p.exp_flags = (int32_t)ip[6];
  p.slope = (float)fp[0]; p.scale = (float)fp[1]; p.out_slope = (float)fp[2];
  DeviceScope scope(c);
  finish(ov_resblock_pair2_bf16cl(&p, c.stream()), "ov_resblock_pair2_bf16cl");
}

// ip = [B, L, Cin, Cout, K, dil, nwg, products, x_plane, out_plane, res_plane, col_limit_scale];
// fp = [res_slope, out_slope, scale]
void conv1d_split3(const OptTensor& x, const OptTensor& w, const OptTensor& bias, const OptTensor& out, const OptTensor& res,
                   const OptTensor& dbg, const OptTensor& col_limit, at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 12 && fp.size() == 3, "openvoice_amd::conv1d_split3: 12 integer and 3 float parameters");
  Ctx c{"conv1d_split3", false};
  ov_conv1d_split3_params p{};
  p.x = sptr<uint16_t>(x, c, 0); p.w = sptr<uint16_t>(w, c, 1); p.bias = sptr<float>(bias, c, 2);
  p.out = sptr<uint16_t>(out, c, 3); p.res = sptr<uint16_t>(res, c, 4);
  p.dbg = sptr<unsigned long long>(dbg, c, 5);
  p.col_limit = sptr<int32_t>(col_limit, c, 6);
  p.B = (int32_t)ip[0]; p.L = (int32_t)ip[1]; p.Cin = (int32_t)ip[2]; p.Cout = (int32_t)ip[3]; p.K = (int32_t)ip[4];
  p.dil = (int32_t)ip[5]; p.nwg = (int32_t)ip[6]; p.products = (int32_t)ip[7];
  p.x_plane = ip[8]; p.out_plane = ip[9]; p.res_plane = ip[10]; p.col_limit_scale = (int32_t)ip[11];
  p.res_slope = (float)fp[0]; p.out_slope = (float)fp[1]; p.scale = (float)fp[2];
  DeviceScope scope(c);
  finish(ov_conv1d_split3(&p, c.stream()), "ov_conv1d_split3");
}

// ip = [B, Cin, Cout, L, x_ld, out_ld, K, dil, nwg, x_bstride, out_bstride, res_bstride, add_bstride, frags,
//       col_limit_scale]; fp = [in_slope, scale, out_slope]
void conv1d_wino_f32(const OptTensor& x, const OptTensor& w, const OptTensor& bias, const OptTensor& out, const OptTensor& res,
                     const OptTensor& add, const OptTensor& dbg, const OptTensor& col_limit, at::IntArrayRef ip,
                     at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 15 && fp.size() == 3, "openvoice_amd::conv1d_wino_f32: 15 integer and 3 float parameters");
  Ctx c{"conv1d_wino_f32", false};
  ov_conv1d_wino_params p{};
  p.x = sptr<float>(x, c, 0); p.w = sptr<float>(w, c, 1); p.bias = sptr<float>(bias, c, 2);
  p.out = sptr<float>(out, c, 3); p.res = sptr<float>(res, c, 4); p.add = sptr<float>(add, c, 5);
  p.dbg = sptr<unsigned long long>(dbg, c, 6);
  p.col_limit = sptr<int32_t>(col_limit, c, 7); p.col_limit_scale = (int32_t)ip[14];
  p.B = (int32_t)ip[0]; p.Cin = (int32_t)ip[1]; p.Cout = (int32_t)ip[2]; p.L = (int32_t)ip[3];
  p.x_ld = (int32_t)ip[4]; p.out_ld = (int32_t)ip[5]; p.K = (int32_t)ip[6]; p.dil = (int32_t)ip[7]; p.nwg = (int32_t)ip[8];
  p.x_bstride = ip[9]; p.out_bstride = ip[10]; p.res_bstride = ip[11]; p.add_bstride = ip[12]; p.frags = (int32_t)ip[13];
  p.in_slope = (float)fp[0]; p.scale = (float)fp[1]; p.out_slope = (float)fp[2];
  DeviceScope scope(c);
  finish(ov_conv1d_wino_f32(&p, c.stream()), "ov_conv1d_wino_f32");
}

}  // namespace

TORCH_LIBRARY(openvoice_amd, m) {
  // ---- parameter-struct entry points
  m.def("conv1d_f32(Tensor? x, Tensor? w, Tensor? bias, Tensor(a!)? out, Tensor? res, Tensor? add, Tensor(b!)? out2, "
        "Tensor? mask, Tensor? bias_b, Tensor? col_limit, int[] ip, float[] fp) -> ()", &conv1d_f32);
  m.def("resblock_pair_f32(Tensor? x, Tensor? w1, Tensor? b1, Tensor? w2, Tensor? b2, Tensor(a!)? out, Tensor? add, "
        "Tensor(b!)? dbg, Tensor? col_limit, int[] ip, float[] fp) -> ()", &resblock_pair_f32);
  m.def("wn_layer_f32(Tensor? x, Tensor(a!)? out, Tensor(b!)? skip, Tensor? w_in, Tensor? b_in, Tensor? cond, Tensor? w_rs, "
        "Tensor? b_rs, Tensor? mask, Tensor(c!)? dbg, Tensor(d!)? acts, int[] ip) -> ()", &wn_layer_f32);
  m.def("conv1d_bf16cl(Tensor? x, Tensor? w, Tensor? bias, Tensor(a!)? out, Tensor? res, Tensor? add, Tensor(b!)? dbg, "
        "int[] ip, float[] fp) -> ()", &conv1d_bf16cl);
  m.def("resblock_pair_bf16cl(Tensor? x, Tensor? w1, Tensor? b1, Tensor? w2, Tensor? b2, Tensor(a!)? out, Tensor? add, "
        "Tensor(b!)? dbg, int[] ip, float[] fp) -> ()", &resblock_pair_bf16cl);
  m.def("resblock_pair2_bf16cl(Tensor? x, Tensor? w1, Tensor? b1, Tensor? w2, Tensor? b2, Tensor(a!)? out, Tensor? add, "
        "Tensor(b!)? dbg, int[] ip, float[] fp) -> ()", &resblock_pair2_bf16cl);
  m.def("conv1d_split3(Tensor? x, Tensor? w, Tensor? bias, Tensor(a!)? out, Tensor? res, Tensor(b!)? dbg, "
        "Tensor? col_limit, int[] ip, float[] fp) -> ()", &conv1d_split3);
  m.def("conv1d_wino_f32(Tensor? x, Tensor? w, Tensor? bias, Tensor(a!)? out, Tensor? res, Tensor? add, Tensor(b!)? dbg, "
        "Tensor? col_limit, int[] ip, float[] fp) -> ()",
        &conv1d_wino_f32);
  // ---- device entry points with flat argument lists (schema derived from the C prototype)
  bind_device<&ov_frame_hops_f32>(m, "frame_hops_f32");
  bind_device<&ov_conv_post_tanh_f32>(m, "conv_post_tanh_f32");
  bind_device<&ov_conv_post_tanh_limited_f32>(m, "conv_post_tanh_limited_f32");
  bind_device<&ov_frame_limits_i32>(m, "frame_limits_i32");
  bind_device<&ov_linear_f32>(m, "linear_f32");
  bind_device<&ov_sequence_mask_f32>(m, "sequence_mask_f32");
  bind_device<&ov_unpad_rows_f32>(m, "unpad_rows_f32");
  bind_device<&ov_polyphase_fir_f32>(m, "polyphase_fir_f32");
  bind_device<&ov_layernorm_freq_f32>(m, "layernorm_freq_f32");
  bind_device<&ov_conv2d_s2_relu_f32>(m, "conv2d_s2_relu_f32");
  bind_device<&ov_gru_f32>(m, "gru_f32");
  bind_device<&ov_embed_f32>(m, "embed_f32");
  bind_device<&ov_layernorm_ch_f32>(m, "layernorm_ch_f32");
  bind_device<&ov_rel_attention_f32>(m, "rel_attention_f32");
  bind_device<&ov_dwconv1d_f32>(m, "dwconv1d_f32");
  bind_device<&ov_expand1_f32>(m, "expand1_f32");
  bind_device<&ov_add_bias_mask_f32>(m, "add_bias_mask_f32");
  bind_device<&ov_rq_spline_inverse_f32>(m, "rq_spline_inverse_f32");
  bind_device<&ov_duration_f32>(m, "duration_f32");
  bind_device<&ov_expand_prior_f32>(m, "expand_prior_f32");
  bind_device<&ov_conv_post_tanh_bf16>(m, "conv_post_tanh_bf16");
  bind_device<&ov_split3_from_f32>(m, "split3_from_f32");
  bind_device<&ov_split3_to_f32>(m, "split3_to_f32");
  // ---- host helpers: weight packers (CPU tensors), sizes, capability queries
  bind_host<&ov_conv1d_pack_f32>(m, "conv1d_pack_f32");
  bind_host<&ov_wn_pack_f32>(m, "wn_pack_f32");
  bind_host<&ov_conv1d_bf16_pack>(m, "conv1d_bf16_pack");
  bind_host<&ov_conv1d_bf16_pack16>(m, "conv1d_bf16_pack16");
  bind_host<&ov_conv1d_split3_pack>(m, "conv1d_split3_pack");
  bind_value<&ov_conv1d_split3_pack_size>(m, "conv1d_split3_pack_size");
  bind_value<&ov_conv1d_split3_supported>(m, "conv1d_split3_supported");
  bind_host<&ov_conv1d_wino_pack_f32>(m, "conv1d_wino_pack_f32");
  bind_value<&ov_conv1d_wino_pack_size>(m, "conv1d_wino_pack_size");
  bind_value<&ov_conv1d_wino_supported>(m, "conv1d_wino_supported");
  bind_value<&ov_conv1d_wino_chunk>(m, "conv1d_wino_chunk");
  bind_value<&ov_conv1d_pack_size>(m, "conv1d_pack_size");
  bind_value<&ov_conv1d_pack_rows>(m, "conv1d_pack_rows");
  bind_value<&ov_wn_pack_size>(m, "wn_pack_size");
  bind_value<&ov_conv1d_bf16_pack_size>(m, "conv1d_bf16_pack_size");
  bind_value<&ov_resblock_pair_supported>(m, "resblock_pair_supported");
  bind_value<&ov_wn_layer_supported>(m, "wn_layer_supported");
  bind_value<&ov_wn_layer_tile>(m, "wn_layer_tile");
  bind_value<&ov_resblock_pair_bf16_supported>(m, "resblock_pair_bf16_supported");
  bind_value<&ov_resblock_pair2_bf16_supported>(m, "resblock_pair2_bf16_supported");
  bind_value<&ov_version>(m, "version");
  bind_value<&ov_build_experiment>(m, "build_experiment");
}
