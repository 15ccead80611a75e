// Thin torch binding of the C ABI ("thin pybind/TORCH_LIBRARY shim converts at::Tensor"):
// `torch.ops.openvoice_amd.*` take tensors, check device / dtype / contiguity with TORCH_CHECK, pick up
// c10::hip::getCurrentHIPStream() and call the extern "C" entry points of libopenvoice_amd.so; a non-zero OV_E_* code
// becomes a RuntimeError.  The library itself stays free of torch types (include/openvoice_amd.h) -- this file is the
// only one that sees both sides.  It is an ALTERNATIVE to the ctypes binding (openvoice_amd/_lib.py), selected with
// OPENVOICE_AMD_BINDING=torch; both launch the same kernels with the same arguments (tests/test_gpu_torch_shim.py).
//
// Built by `make -C openvoice_amd/csrc torch_shim` (hipcc, in-tree: openvoice_amd/libopenvoice_amd_torch.so).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

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

void check_f32(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, ": must be on a ROCm device (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, ": must be float32");
}
const float* fptr(const c10::optional<at::Tensor>& t, int64_t off, const char* name) {
  if (!t.has_value() || !t->defined()) return nullptr;
  check_f32(*t, name);
  return t->data_ptr<float>() + off;
}
ov_stream_t current_stream(const at::Tensor& on) {
  return static_cast<ov_stream_t>(c10::hip::getCurrentHIPStream(on.device().index()).stream());
}

// ip = [B, Cin, L, x_ld, out_ld, M, Cout, K, dil, epi, flags, split, phase_s, tiles_per_wg, tile, loaders, chunk,
//       x_bstride, out_bstride, res_bstride, add_bstride, out2_bstride, bias_b_bstride, mask_bstride,
//       x_off, out_off, res_off, bias_b_off]   (strides and offsets in elements);  fp = [in_slope, scale]
void conv1d(const at::Tensor& x, const at::Tensor& w, const at::Tensor& bias, at::Tensor out,
            const c10::optional<at::Tensor>& res, const c10::optional<at::Tensor>& add,
            const c10::optional<at::Tensor>& out2, const c10::optional<at::Tensor>& mask,
            const c10::optional<at::Tensor>& bias_b, at::IntArrayRef ip, at::ArrayRef<double> fp) {
  TORCH_CHECK(ip.size() == 28 && fp.size() == 2, "openvoice_amd::conv1d: 28 integer and 2 float parameters");
  check_f32(x, "x"); check_f32(w, "w"); check_f32(bias, "bias"); check_f32(out, "out");
  TORCH_CHECK(x.device() == out.device() && x.device() == w.device(), "conv1d: tensors on different devices");
  ov_conv1d_params p{};
  p.x = x.data_ptr<float>() + ip[24];
  p.w = w.data_ptr<float>();
  p.bias = bias.data_ptr<float>();
  p.out = out.data_ptr<float>() + ip[25];
  p.res = fptr(res, ip[26], "res");
  p.add = fptr(add, 0, "add");
  p.out2 = const_cast<float*>(fptr(out2, 0, "out2"));
  p.mask = fptr(mask, 0, "mask");
  p.bias_b = fptr(bias_b, ip[27], "bias_b");
  p.B = (int32_t)ip[0]; p.Cin = (int32_t)ip[1]; p.L = (int32_t)ip[2]; p.x_ld = (int32_t)ip[3]; p.out_ld = (int32_t)ip[4];
  p.M = (int32_t)ip[5]; p.Cout = (int32_t)ip[6]; p.K = (int32_t)ip[7]; p.dil = (int32_t)ip[8]; p.epi = (int32_t)ip[9];
  p.flags = (int32_t)ip[10]; p.split = (int32_t)ip[11]; p.phase_s = (int32_t)ip[12]; p.tiles_per_wg = (int32_t)ip[13];
  p.tile = (int32_t)ip[14]; p.loaders = (int32_t)ip[15]; p.chunk = (int32_t)ip[16];
  p.x_bstride = ip[17]; p.out_bstride = ip[18]; p.res_bstride = ip[19]; p.add_bstride = ip[20];
  p.out2_bstride = ip[21]; p.bias_b_bstride = ip[22]; p.mask_bstride = ip[23];
  p.in_slope = (float)fp[0]; p.scale = (float)fp[1];
  const int rc = ov_conv1d_f32(&p, current_stream(x));
  TORCH_CHECK(rc == OV_OK, "ov_conv1d_f32 failed: ", ov_strerror(rc));
}

void resblock_pair(const at::Tensor& x, const at::Tensor& w1, const at::Tensor& b1, const at::Tensor& w2,
                   const at::Tensor& b2, at::Tensor out, const c10::optional<at::Tensor>& add, int64_t B, int64_t C,
                   int64_t L, int64_t ld, int64_t K, int64_t dil, int64_t x_bstride, int64_t out_bstride,
                   int64_t add_bstride, double slope, double scale) {
  check_f32(x, "x"); check_f32(w1, "w1"); check_f32(b1, "b1"); check_f32(w2, "w2"); check_f32(b2, "b2"); check_f32(out, "out");
  ov_respair_params p{};
  p.x = x.data_ptr<float>(); p.w1 = w1.data_ptr<float>(); p.b1 = b1.data_ptr<float>();
  p.w2 = w2.data_ptr<float>(); p.b2 = b2.data_ptr<float>(); p.out = out.data_ptr<float>();
  p.add = fptr(add, 0, "add");
  p.x_bstride = x_bstride; p.out_bstride = out_bstride; p.add_bstride = add_bstride;
  p.B = (int32_t)B; p.C = (int32_t)C; p.L = (int32_t)L; p.ld = (int32_t)ld; p.K = (int32_t)K; p.dil = (int32_t)dil;
  p.slope = (float)slope; p.scale = (float)scale;
  const int rc = ov_resblock_pair_f32(&p, current_stream(x));
  TORCH_CHECK(rc == OV_OK, "ov_resblock_pair_f32 failed: ", ov_strerror(rc));
}

void conv_post_tanh(const at::Tensor& x, const at::Tensor& w, at::Tensor out, int64_t B, int64_t C, int64_t L, int64_t K,
                    double in_slope) {
  check_f32(x, "x"); check_f32(w, "w"); check_f32(out, "out");
  TORCH_CHECK(x.is_contiguous() && out.is_contiguous() && w.is_contiguous(), "conv_post_tanh: contiguous tensors");
  const int rc = ov_conv_post_tanh_f32(x.data_ptr<float>(), w.data_ptr<float>(), out.data_ptr<float>(), (int)B, (int)C,
                                       (int)L, (int)K, (float)in_slope, current_stream(x));
  TORCH_CHECK(rc == OV_OK, "ov_conv_post_tanh_f32 failed: ", ov_strerror(rc));
}

at::Tensor linear(const at::Tensor& x, const at::Tensor& w, const at::Tensor& bias) {
  check_f32(x, "x"); check_f32(w, "w"); check_f32(bias, "bias");
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.size(1) == w.size(1) && bias.numel() == w.size(0), "linear: shapes");
  TORCH_CHECK(x.is_contiguous() && w.is_contiguous() && bias.is_contiguous(), "linear: contiguous tensors");
  at::Tensor y = at::empty({x.size(0), w.size(0)}, x.options());
  const int rc = ov_linear_f32(x.data_ptr<float>(), w.data_ptr<float>(), bias.data_ptr<float>(), y.data_ptr<float>(),
                               (int)x.size(0), (int)w.size(0), (int)x.size(1), current_stream(x));
  TORCH_CHECK(rc == OV_OK, "ov_linear_f32 failed: ", ov_strerror(rc));
  return y;
}

void sequence_mask(const at::Tensor& lengths, at::Tensor mask, int64_t B, int64_t T, int64_t ld) {
  TORCH_CHECK(lengths.is_cuda() && lengths.scalar_type() == at::kLong && lengths.is_contiguous(), "lengths: int64 on device");
  check_f32(mask, "mask");
  const int rc = ov_sequence_mask_f32(lengths.data_ptr<int64_t>(), mask.data_ptr<float>(), (int)B, (int)T, (int)ld,
                                      current_stream(mask));
  TORCH_CHECK(rc == OV_OK, "ov_sequence_mask_f32 failed: ", ov_strerror(rc));
}

int64_t version() { return ov_version(); }

}  // namespace

TORCH_LIBRARY(openvoice_amd, m) {
  m.def("conv1d(Tensor x, Tensor w, Tensor bias, Tensor(a!) out, Tensor? res, Tensor? add, Tensor(b!)? out2, Tensor? mask, "
        "Tensor? bias_b, int[] ip, float[] fp) -> ()", &conv1d);
  m.def("resblock_pair(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor(a!) out, Tensor? add, int B, int C, "
        "int L, int ld, int K, int dil, int x_bstride, int out_bstride, int add_bstride, float slope, float scale) -> ()",
        &resblock_pair);
  m.def("conv_post_tanh(Tensor x, Tensor w, Tensor(a!) out, int B, int C, int L, int K, float in_slope) -> ()", &conv_post_tanh);
  m.def("linear(Tensor x, Tensor w, Tensor bias) -> Tensor", &linear);
  m.def("sequence_mask(Tensor lengths, Tensor(a!) mask, int B, int T, int ld) -> ()", &sequence_mask);
  m.def("version() -> int", &version);
}
