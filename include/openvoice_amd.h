/*
 * openvoice_amd.h -- C ABI of the MI355X (gfx950) tone-colour-converter kernels.
 *
 * The reference (myshell-ai/OpenVoice) is pure Python/PyTorch and has no FFI of its own; the
 * seam these entry points replace is the ATen calls issued by
 * SynthesizerTrn.voice_conversion (reference: openvoice/models.py:492-499) and
 * SynthesizerTrn.ref_enc (openvoice/models.py:339-359).  Each function below names the
 * reference lines whose arithmetic it implements.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - All tensors are fp32, contiguous along time, layout (B, C, T); pointers are DEVICE
 *    pointers unless the parameter says "host".  Caller owns every buffer; nothing here
 *    allocates, frees or synchronises.  Launches go to `stream` and return immediately.
 *  - Return value: 0 = OV_OK, negative = OV_E_* (nothing was launched).
 *  - Weights are consumed in a packed, MFMA-fragment-ordered layout produced once at load
 *    time by ov_conv1d_pack_f32 (host function).
 */
#ifndef OPENVOICE_AMD_H
#define OPENVOICE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ov_stream_t; /* hipStream_t */

enum {
  OV_OK = 0,
  OV_E_BADARG = -1,      /* null pointer / non-positive size / inconsistent params */
  OV_E_UNSUPPORTED = -2, /* (kernel size, dilation, tile) combination not instantiated */
  OV_E_ALIGN = -3,       /* pointer/stride alignment the kernel needs is not met */
  OV_E_LAUNCH = -4       /* hipGetLastError() != hipSuccess after the launch */
};

/* Epilogues of the implicit-GEMM Conv1d kernel (v = acc + bias[row] + bias_b[b][row]). */
enum {
  /* out[b][row][t] = ((v [*mask]) [+ res] [+ add]) * scale                                   */
  OV_EPI_LINEAR = 0,
  /* WaveNet gate, reference openvoice/commons.py:100-107 via modules.py:200:
   * out[b][c][t] = tanh(v[2 tiles paired: row c]) * sigmoid(v[row c + H]); rows are packed so
   * that 32-row tile 2q holds tanh rows 32q.. and tile 2q+1 holds sigmoid rows H+32q..        */
  OV_EPI_GATE = 1,
  /* WaveNet res/skip, reference openvoice/modules.py:203-209:
   * row < split : out[b][row][t]  = (out[b][row][t] + v) * mask[b][t]      (residual, in place)
   * row >= split: out2[b][row-split][t] (+)= v     ('=' when OV_F_OUT2_INIT, else '+=')      */
  OV_EPI_RESSKIP = 2,
  /* mean-only coupling combine, reference openvoice/modules.py:441-455 (out is x1, in place):
   * forward (scale > 0): out = v*mask + out*mask ;  reverse (scale < 0): out = (out - v*mask)*mask */
  OV_EPI_COUPLE = 3,
  /* posterior sample, reference openvoice/models.py:218-220 (rows paired like OV_EPI_GATE:
   * m = tile 2q, logs = tile 2q+1): out = (m*mask + res*scale*exp(logs*mask))*mask,
   * res = noise, scale = tau                                                                   */
  OV_EPI_POSTERIOR = 4,
  /* ConvTranspose1d written as a 3-tap conv over stride-many output phases, reference
   * openvoice/models.py:279 (weights packed with row = cout*phase_s + phase):
   * out[b][cout][phase_s*t + phase] = v                                                        */
  OV_EPI_CONVT = 5
};

enum {
  OV_F_MASK_V = 1,    /* LINEAR: multiply v by mask[b][t] before the residual add */
  OV_F_OUT2_INIT = 2  /* RESSKIP: out2 = v instead of out2 += v (first WaveNet layer) */
};

/* One Conv1d launch.  Input length == output length L ('same' padding, stride 1), as every conv
 * on the converter path (reference: openvoice/modules.py:163-171, :228-283, models.py:238,266).
 * Rows may be padded: x rows are x_ld floats apart, out/res/add/out2 rows out_ld floats apart (both
 * >= L; columns >= L are never read as data and never written).  16-byte staging loads are used
 * when x, x_bstride and x_ld are 16-byte aligned -- L itself may be ragged. */
typedef struct ov_conv1d_params {
  const float* x;        /* [B][>=Cin][L]; channel offset already applied to the pointer        */
  const float* w;        /* packed weights from ov_conv1d_pack_f32                              */
  const float* bias;     /* [M rounded up to 128] in packed row order; required (zeros if none) */
  const float* bias_b;   /* per-batch bias [B or 1][M] in packed row order, or NULL             */
  float* out;            /* primary output, indexed [b][row][t] with out_bstride                */
  const float* res;      /* LINEAR: residual; POSTERIOR: noise; indexed like out; or NULL       */
  const float* add;      /* LINEAR: second addend (MRF running sum), indexed like out; or NULL  */
  float* out2;           /* RESSKIP: skip accumulator [b][row-split][t]                         */
  const float* mask;     /* [B][>=L] sequence mask (1/0), rows mask_bstride apart, or NULL      */
  int64_t x_bstride;     /* elements between consecutive batches of x                           */
  int64_t out_bstride;
  int64_t res_bstride;
  int64_t add_bstride;
  int64_t out2_bstride;
  int64_t bias_b_bstride;/* 0 broadcasts one row to every batch item                            */
  int64_t mask_bstride;  /* 0 = L                                                               */
  int32_t B, Cin, L;
  int32_t x_ld;          /* row stride of x in floats; 0 = L                                    */
  int32_t out_ld;        /* row stride of out/res/add/out2; 0 = L (CONVT: L * phase_s)          */
  int32_t M;             /* packed rows = 32 * m_tiles, as passed to ov_conv1d_pack_f32          */
  int32_t Cout;          /* rows >= Cout (LINEAR/COUPLE/RESSKIP) are padding and never stored    */
  int32_t K, dil;        /* taps, dilation; padding is (K-1)*dil/2                               */
  int32_t epi, flags, split, phase_s;
  int32_t tiles_per_wg;  /* consecutive time tiles walked by one workgroup; 0 = 1                */
  int32_t tile;          /* 0 = chosen by the dispatcher; else 1 + tile id (128x128, 64x256,
                          * 32x512, 32x256) -- tuning / measurement knob                         */
  int32_t loaders;       /* loader waves per workgroup: 0 = chosen by the dispatcher, else 1/2/4 */
  int32_t chunk;         /* input channels per LDS fill: 0 = default (32 for 1x1, else 16), or 16/32 */
  float in_slope;        /* leaky-ReLU slope applied to x while staging (1.0f = identity)        */
  float scale;
} ov_conv1d_params;

/* ---- host-side helpers -------------------------------------------------------------------- */

/* Number of floats ov_conv1d_pack_f32 writes for a dense [Cout][Cin][K] weight. */
size_t ov_conv1d_pack_size(int Cout, int Cin, int K);
/* Rows of the packed weight (Cout rounded up to the packing granule, 128). */
int ov_conv1d_pack_rows(int Cout);
/* Re-lay a dense row-major HOST weight w[Cout][Cin][K] into MFMA A-fragment order (HOST dst).
 * Replaces nothing in the reference; it is the load-time transform that lets the kernel read
 * every weight fragment as one coalesced 1 KiB record.  Layout: DESIGN.md "Packed weights". */
int ov_conv1d_pack_f32(const float* w, int Cout, int Cin, int K, float* dst);

/* ---- device entry points ------------------------------------------------------------------ */

/* Implicit-GEMM Conv1d on fp32 MFMA with fused prologue/epilogue.  Replaces F.conv1d /
 * F.conv_transpose1d + the surrounding elementwise ops at: openvoice/modules.py:194-209 (WN),
 * :296-306 (ResBlock1), :439-455 (coupling), openvoice/models.py:216-220 (posterior encoder),
 * :273-286 (generator conv_pre, ups, MRF). */
int ov_conv1d_f32(const ov_conv1d_params* p, ov_stream_t stream);

/* conv_post + tanh, reference openvoice/models.py:287-289:
 * out[b][0][t] = tanh( sum_{c,j} w[c][j] * lrelu(x[b][c][t+j-(K-1)/2], in_slope) ), no bias.
 * w is the dense [C][K] DEVICE weight. */
int ov_conv_post_tanh_f32(const float* x, const float* w, float* out, int B, int C, int L, int K,
                          float in_slope, ov_stream_t stream);

/* y[b][m] = bias[m] + sum_k w[m][k] * x[b][k]  (dense row-major w, all DEVICE).  The T=1
 * conditioning convs, reference openvoice/modules.py:189-190 (cond_layer) and
 * openvoice/models.py:275 (dec.cond), and ref_enc.proj (models.py:359). */
int ov_linear_f32(const float* x, const float* w, const float* bias, float* y, int B, int M, int Kdim,
                  ov_stream_t stream);

/* mask[b][t] = t < lengths[b] ? 1 : 0 for t < T, reference openvoice/commons.py:121-125.
 * Rows of mask are ld floats apart (0 = T); columns >= T are not written. */
int ov_sequence_mask_f32(const int64_t* lengths, float* mask, int B, int T, int ld, ov_stream_t stream);

/* ---- ReferenceEncoder (extract_se), reference openvoice/models.py:339-359 -----------------------
 * Layout: time is the contiguous axis everywhere, [N][C][F][T]; the spectrogram [N][F][T] is the
 * C = 1 case, and the conv-stack output [N][128][9][T'] read as [N][1152][T'] is the GRU input in
 * the (B, C, T) layout of ov_conv1d_f32 (feature index c*9 + f, as models.py:351-354 builds it). */

/* y[n][f][t] = LayerNorm over f of x[n][:, t] (nn.LayerNorm(F), reference models.py:344). */
int ov_layernorm_freq_f32(const float* x, const float* gamma, const float* beta, float* y, int N, int F, int T,
                          float eps, ov_stream_t stream);

/* y = relu(conv2d(x, w, bias, stride 2, pad 1)), 3x3; w is the reference's [Cout][Cin][kh=time][kw=freq]
 * (models.py:314-325, :346-349).  x [N][Cin][Fi][Ti] -> y [N][Cout][(Fi-1)/2+1][(Ti-1)/2+1].
 * Cout must be a multiple of 16. */
int ov_conv2d_s2_relu_f32(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                          int Fi, int Ti, ov_stream_t stream);

/* GRU recurrence over T steps from h0 = 0, final hidden state out (nn.GRU batch_first, models.py:356-357).
 * gi [N][3H][T] = W_ih x_t + b_ih (gate order r,z,n), whh_t = W_hh transposed to [H][3H], bhh [3H],
 * h_out [N][H].  H must be 128. */
int ov_gru_f32(const float* gi, const float* whh_t, const float* bhh, float* h_out, int N, int H, int T,
               ov_stream_t stream);

/* Library/ABI version (major*100 + minor). */
int ov_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OPENVOICE_AMD_H */
