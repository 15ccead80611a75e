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
  /* mean-only coupling combine, reference openvoice/modules.py:441-455; x1 = res when res != NULL (indexed like
   * out), else out itself (in place):
   * forward (scale > 0): out = v*mask + x1*mask ;  reverse (scale < 0): out = (x1 - v*mask)*mask */
  OV_EPI_COUPLE = 3,
  /* posterior sample, reference openvoice/models.py:218-220 (rows paired like OV_EPI_GATE:
   * m = tile 2q, logs = tile 2q+1): out = (m*mask + res*scale*exp(logs*mask))*mask,
   * res = noise, scale = tau                                                                   */
  OV_EPI_POSTERIOR = 4,
  /* ConvTranspose1d (k = 2*phase_s, padding phase_s/2) written as a 3-tap conv over stride-many output phases,
   * reference openvoice/models.py:279 (weights packed with row = cout*phase_s + phase):
   * out[b][cout][phase_s*t + phase] = v.
   * Every phase has exactly two non-zero taps: x[t-1], x[t] for phase < phase_s/2 and x[t], x[t+1] above.  With
   * OV_F_CONVT_GROUPED (phase_s 8 or 2 only) the rows are packed by phase group instead -- 32-row tile 2q = the
   * phases < phase_s/2, tile 2q+1 the phases >= phase_s/2 of the same output channels (phase_s 8: row-in-tile =
   * 4*(cout % 8) + phase % 4 for channels 8q..8q+7; phase_s 2: row-in-tile = cout % 32 for channels 32q..32q+31)
   * -- and the kernel skips the all-zero tap of each tile: 2/3 of the matrix work.                          */
  OV_EPI_CONVT = 5,
  /* Spectrogram magnitude, reference openvoice/mel_processing.py:61-74: rows paired like OV_EPI_GATE
   * (tile 2q = real parts of bins 32q.., tile 2q+1 = imaginary parts): out[b][f][t] = sqrt(re^2 + im^2 + scale)
   * for f < Cout (Cout need not be a multiple of 32); no bias.  Used with the even-K framing conv.           */
  OV_EPI_MAGNITUDE = 6
};

enum {
  OV_F_MASK_V = 1,    /* LINEAR: multiply v by mask[b][t] before the residual add */
  OV_F_OUT2_INIT = 2, /* RESSKIP: out2 = v instead of out2 += v (first WaveNet layer) */
  OV_F_CONVT_GROUPED = 4, /* CONVT, phase_s 8 / 2: rows packed by phase group (see OV_EPI_CONVT); M % 64 == 0 */
  OV_F_NO_XCD_MAP = 8 /* measurement knob: walk the tile list round-robin over the workgroups instead of giving each
                       * XCD (workgroup id % 8) a contiguous eighth of it; results are identical either way */
};

/* One Conv1d launch.  Input length == output length L ('same' padding, stride 1), as every conv
 * on the converter path (reference: openvoice/modules.py:163-171, :228-283, models.py:238,266).
 * Rows may be padded: x rows are x_ld floats apart, out/res/add/out2 rows out_ld floats apart (both
 * >= L; columns >= L are never read as data and never written).  The forward-aligned even-K conv reads
 * L + (K-1)*dil input columns (x_ld must cover them) and writes L.  16-byte staging loads are used
 * when x, x_bstride and x_ld are 16-byte aligned -- L itself may be ragged. */
typedef struct ov_conv1d_params {
  const float* x;        /* [B][>=Cin][L]; channel offset already applied to the pointer        */
  const float* w;        /* packed weights from ov_conv1d_pack_f32                              */
  const float* bias;     /* [M rounded up to 128] in packed row order; required (zeros if none) */
  const float* bias_b;   /* per-batch bias [B or 1][M] in packed row order, or NULL             */
  float* out;            /* primary output, indexed [b][row][t] with out_bstride                */
  const float* res;      /* LINEAR: residual; POSTERIOR: noise; COUPLE: x1 source; indexed like out; or NULL */
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
  int32_t K, dil;        /* taps, dilation; odd K: 'same' padding (K-1)*dil/2; even K: taps t .. t+(K-1)*dil,
                          * no left padding (the framing conv of the spectrogram)                   */
  int32_t epi, flags, split, phase_s;
  int32_t tiles_per_wg;  /* 0 = the dispatcher's launch rule (persistent -- one workgroup per resident slot, striding
                          * over the tiles -- for large launches, else one tile per workgroup);
                          * n > 0 = ceil(tiles / n) workgroups; n < 0 = persistent, forced -- tests / measurement */
  int32_t tile;          /* 0 = chosen by the dispatcher; else 1 + tile id (128x128, 64x256,
                          * 32x512, 32x256) -- tuning / measurement knob                         */
  int32_t loaders;       /* loader waves per workgroup: 0 = chosen by the dispatcher, else 1/2/4            */
  int32_t chunk;         /* input channels per LDS fill: 0 = default (32 for 1x1, else 16), or 16/32 */
  float in_slope;        /* leaky-ReLU slope applied to x while staging (1.0f = identity)        */
  float scale;
  /* Length-aware work list (padded batches: ragged convert_batch, batched TTS -- reference openvoice/models.py:477-489
   * pads every utterance to the longest).  col_limit[b] * col_limit_scale = the output columns of utterance b that
   * matter (clamped to [0, L]); time tiles wholly beyond it are dropped from the work list -- their columns of out
   * are NOT written -- and the remaining tiles are dealt evenly over the workgroups.  Every computed value is
   * bit-identical to the full launch.  NULL = the whole tensor.  DEVICE pointer, int32 [B]; ignored (whole tensor)
   * when B > 256.  The caller owns the margin: the generator's receptive field is 13.3 frames, so a limit of
   * length + 16 frames leaves every valid sample unchanged (ov_frame_limits_i32). */
  const int32_t* col_limit;
  int32_t col_limit_scale; /* columns of this launch per unit of col_limit (e.g. the stage's upsampling factor) */
  int32_t reserved0;
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

/* One ResBlock1 iteration in ONE launch, reference openvoice/modules.py:296-306 (the loop body of
 * ResBlock1.forward with x_mask = None, as the generator calls it at models.py:280-286):
 *   out = ( c2( lrelu( c1( lrelu(x, slope) ), slope ) ) + x [+ add] ) * scale
 * c1 = Conv1d(C, C, K, dilation dil), c2 = Conv1d(C, C, K, dilation 1), both 'same'-padded; w1 / w2 packed by
 * ov_conv1d_pack_f32(C, C, K), b1 / b2 [128] in packed row order.  `add` / `scale` carry the MRF running sum and
 * the final 1/num_kernels of models.py:282-286.  The intermediate tensor stays in LDS (sliding window along time,
 * openvoice_amd/csrc/conv1d_pair.h); results are bit-identical to the two ov_conv1d_f32 launches it replaces.
 * x / out / add are [B][C][L] with rows `ld` floats apart (0 = L), 16-byte aligned rows (ld % 4 == 0);
 * out must not alias x or add.  Instantiated for the HBM-bound stages only: ov_resblock_pair_supported(). */
typedef struct ov_respair_params {
  const float* x;
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  float* out;
  const float* add;      /* or NULL */
  int64_t x_bstride, out_bstride, add_bstride;
  int32_t B, C, L;
  int32_t ld;            /* row stride of x / out / add in floats; 0 = L */
  int32_t K, dil;
  int32_t nwg;           /* 0 = one workgroup per resident slot; n > 0 forces n workgroups (tests) */
  float slope;           /* leaky-ReLU slope of both activations (modules.py:14: 0.1) */
  float scale;
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][4 waves][8] shader-clock ticks per
                          * phase (residual issue, chunk wait, c1, h store, h barrier, c2, epilogue, tail) */
  const int32_t* col_limit; /* length-aware work list, as ov_conv1d_params.col_limit: utterance b is walked only up
                          * to column col_limit[b] * col_limit_scale (steps beyond are neither computed nor written;
                          * what is computed is bit-identical to the full launch); NULL = whole tensor; DEVICE int32 [B],
                          * B <= 256 */
  int32_t col_limit_scale;
  int32_t reserved0;
} ov_respair_params;
int ov_resblock_pair_f32(const ov_respair_params* p, ov_stream_t stream);
/* 1 when (C, K, dil) has a fused instance, else 0 (callers then issue the two ov_conv1d_f32 launches). */
int ov_resblock_pair_supported(int C, int K, int dil);

/* One WaveNet layer in ONE launch, reference openvoice/modules.py:192-209 (the loop body of WN.forward) with
 * commons.py:100-107 (fused_add_tanh_sigmoid_multiply):
 *   x_in = in_layer(x) + cond;  acts = tanh(x_in[:H]) * sigmoid(x_in[H:]);  rs = res_skip_layer(acts)
 *   out  = (x + rs[:H]) * mask;  skip = (first ? 0 : skip) + rs[H:]       (last layer: rs has H rows, all skip)
 * x / out / skip are [B][H][T] with rows ld floats apart and batch items bstride apart; out must not alias x (tiles
 * read a (K-1)/2-column halo of x from their neighbours: the caller ping-pongs two buffers).  `acts` never leaves
 * the CU (openvoice_amd/csrc/wn_layer.hip).
 * Weights: w_in = ov_wn_pack_f32 of the dense [2H][H][K] in_layer weight with its ROWS in gate order -- 16-row
 * block q holds, at row 4j + {0, 1, 2, 3}, the tanh rows of channels 8q + j and 8q + j + 4 followed by their
 * sigmoid rows (j = 0..3); b_in [2H] and cond [B or 1][2H] (cond_bstride 0 broadcasts) in the same row order.
 * w_rs = ov_wn_pack_f32 of the dense [2H][H][1] res_skip weight in natural row order (last layer: rows 0..H-1 zero,
 * the H skip rows at H..2H-1), b_rs [2H] likewise.
 * Replaces two ov_conv1d_f32 launches (OV_EPI_GATE, OV_EPI_RESSKIP); results agree with them to fp32 rounding
 * (different summation order; the gate uses v_exp_f32 / v_rcp_f32 instead of libm tanhf / expf: <= 4e-7 absolute). */
typedef struct ov_wn_layer_params {
  const float* x;
  float* out;
  float* skip;
  const float* w_in;
  const float* b_in;
  const float* cond;     /* or NULL */
  const float* w_rs;
  const float* b_rs;
  const float* mask;     /* [B][T] 0/1, rows mask_bstride apart (0 = ld); 16-byte aligned, mask_bstride a multiple
                          * of 4 and >= T rounded up to 4 (read as 16-byte vectors: OV_E_ALIGN / OV_E_BADARG otherwise) */
  int64_t bstride;       /* batch stride of x / out / skip */
  int64_t cond_bstride;
  int64_t mask_bstride;
  int32_t B, H, T;
  int32_t ld;            /* row stride in floats, multiple of 4; 0 = T */
  int32_t K;
  int32_t first;         /* 1: skip is initialised, not accumulated (layer 0) */
  int32_t last;          /* 1: the layer has no residual rows; out is not written */
  int32_t width;         /* tile width in columns: 0 = chosen by the launcher (ov_wn_layer_tile), else 16 .. 128 step 16 */
  int32_t ntile;         /* filled in by the launcher */
  int32_t row_split;     /* 0: the launcher decides; 1: always the fused launch; 3: always the row-split pair (needs acts,
                          * dbg must be NULL) */
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][8 matrix waves][8] shader-clock ticks
                          * per phase (first chunk wait, gate-conv k-steps, chunk waits, gate, operand issue + acts
                          * barrier, res/skip k-steps, epilogue) and the wave's start tick */
  float* acts;           /* ABI 2.05; or NULL: scratch [B][H][ld] (batch stride = bstride, 16-byte aligned, not x / out /
                          * skip).  With it the launcher may run the layer as TWO launches whose workgroups each own a
                          * third of the rows -- gate rows -> acts, then res/skip rows -- when B * ceil(T / 16) tiles would
                          * leave most compute units idle (one or two utterances at frame rate); results bit-identical */
} ov_wn_layer_params;
int ov_wn_layer_f32(const ov_wn_layer_params* p, ov_stream_t stream);
/* 1 when (hidden channels, taps) has a fused instance (192, 5: every WN of the converter and of the V1 speaker). */
int ov_wn_layer_supported(int H, int K);
/* Floats written by ov_wn_pack_f32 (0 when rows % 128 or cin % 32). */
size_t ov_wn_pack_size(int rows, int cin, int K);
/* Dense HOST w[rows][cin][K] -> 16x16x4 A-fragment order, [wave][record][row block][lane][4 k-steps] (HOST dst). */
int ov_wn_pack_f32(const float* w, int rows, int cin, int K, float* dst);
/* Tile width the launcher uses for (B, T) on the current device: equal tiles per utterance, a multiple of 16
 * columns, that fill the compute units in whole rounds (width > 0: validated and returned; 0 = invalid). */
int ov_wn_layer_tile(int B, int T, int width);

/* Framing for the linear spectrogram, reference openvoice/mel_processing.py:54-58 (reflect pad) and the framing
 * step of torch.stft at :61-72: hops[b][c][u] = ypad[hop*u + c] for u < U, with ypad the waveform [B][N]
 * reflect-padded by `pad` samples on each side (zero beyond that).  hops is (B, hop, U) with rows ld apart.
 * With n_fft = 4*hop the windowed DFT + magnitude is then ov_conv1d_f32 with K = 4, Cin = hop and
 * OV_EPI_MAGNITUDE (weights = window * cos / -sin, rows paired re/im). */
int ov_frame_hops_f32(const float* wave, float* hops, int B, int N, int hop, int pad, int U, int ld,
                      ov_stream_t stream);

/* Rate conversion at the audio boundary, reference openvoice/api.py:123,144 (``librosa.load(path, sr=...)`` = resampy's
 * kaiser_best band-limited sinc interpolation): a polyphase FIR over a mono waveform,
 *   y[t] = sum_{j < 2 taps} h[t % P][j] * x[(t * Q) / P - taps + 1 + j]        (x = 0 outside [0, n_in))
 * with P / Q = output rate / input rate in lowest terms and h the [P][2 taps] float64 weights of the interpolation
 * filter at each of the P fractional positions (openvoice_amd/audio_io.py: kaiser_best_phases, the same weights its host
 * restatement applies); float64 accumulation.  All pointers DEVICE.  ABI 2.06. */
int ov_polyphase_fir_f32(const float* x, const double* h, float* y, int64_t n_in, int64_t n_out, int P, int Q, int taps,
                         ov_stream_t stream);

/* conv_post + tanh, reference openvoice/models.py:287-289:
 * out[b][0][t] = tanh( sum_{c,j} w[c][j] * lrelu(x[b][c][t+j-(K-1)/2], in_slope) ), no bias.
 * w is the dense [C][K] DEVICE weight. */
int ov_conv_post_tanh_f32(const float* x, const float* w, float* out, int B, int C, int L, int K,
                          float in_slope, ov_stream_t stream);
/* The same with a length-aware tail: samples t >= col_limit[b] * col_limit_scale of utterance b are written as 0
 * without reading x there (the generator launches before it left those columns unwritten, see
 * ov_conv1d_params.col_limit).  col_limit NULL = ov_conv_post_tanh_f32.  (A col_limit_scale that is not a multiple
 * of 4 takes the per-sample kernel: the 16-byte path decides per 4 consecutive samples.) */
int ov_conv_post_tanh_limited_f32(const float* x, const float* w, float* out, int B, int C, int L, int K,
                                  float in_slope, const int32_t* col_limit, int col_limit_scale, ov_stream_t stream);
/* limits[b] = min(T, max(0, lengths[b]) + margin): the frames of utterance b the generator has to produce so that
 * every sample of its first lengths[b] frames is unaffected by what lies beyond (margin >= 14: the generator's
 * one-sided receptive field is 13.3 frames -- conv_pre 3, ups 1 + 1/8 + 1/64 + 1/128, MRF 60 samples per stage,
 * conv_post 3 samples; reference openvoice/models.py:225-291). */
int ov_frame_limits_i32(const int64_t* lengths, int32_t* limits, int B, int T, int margin, ov_stream_t stream);

/* y[b][m] = bias[m] + sum_k w[m][k] * x[b][k]  (dense row-major w, all DEVICE).  The T=1
 * conditioning convs, reference openvoice/modules.py:189-190 (cond_layer) and
 * openvoice/models.py:275 (dec.cond), and ref_enc.proj (models.py:359). */
int ov_linear_f32(const float* x, const float* w, const float* bias, float* y, int B, int M, int Kdim,
                  ov_stream_t stream);

/* mask[b][t] = t < lengths[b] ? 1 : 0 for t < T, reference openvoice/commons.py:121-125.
 * Rows of mask are ld floats apart (0 = T); columns >= T are not written. */
int ov_sequence_mask_f32(const int64_t* lengths, float* mask, int B, int T, int ld, ov_stream_t stream);
/* dst[r][t] = src[r * ld + t], t < T: the engine's 16-byte aligned rows -> the dense [.., T] tensors the model seam
 * returns (z, z_p, z_hat, y_mask of openvoice/models.py:499). */
int ov_unpad_rows_f32(const float* src, float* dst, int rows, int T, int ld, ov_stream_t stream);

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

/* ---- V1 base-speaker TTS front end (SynthesizerTrn.infer, reference openvoice/models.py:467-490) ----
 * Token-rate tensors are (B, C, T) fp32 with rows `ld` floats apart (ld >= T), like the frame-rate
 * tensors of the converter.  The dense 1x1 / k3 convs of this path go through ov_conv1d_f32. */

/* out[b][h][t] = t < lengths[b] ? emb[tokens[b][t]][h] * scale : 0 -- nn.Embedding * sqrt(H), transposed and
 * masked (models.py:49-54).  tokens [B][T] int64 (ids outside [0, V) are clamped; check them on the host). */
int ov_embed_f32(const int64_t* tokens, const float* emb, const int64_t* lengths, float* out, int B, int T, int H,
                 int V, int ld, float scale, ov_stream_t stream);

enum {
  OV_LN_PRE_RELU = 1, /* relu before the statistics: DurationPredictor conv -> relu -> norm (models.py:91-97) */
  OV_LN_POST_GELU = 2 /* exact (erf) GELU after the affine: DDSConv (modules.py:122-126)                      */
};
/* Channel LayerNorm of modules.py:17-29 / attentions.py:12-24 with the surrounding elementwise ops fused:
 * v = x (+ res) [relu]; y = (v - mean_c) * rstd_c * gamma + beta [gelu]; y += res2; y *= mask[b][t].
 * res, res2, mask may be NULL; out may alias x or res2. */
int ov_layernorm_ch_f32(const float* x, const float* res, const float* gamma, const float* beta, const float* res2,
                        const float* mask, float* out, int B, int C, int T, int ld, float eps, int flags,
                        ov_stream_t stream);

/* Multi-head self-attention with windowed relative-position keys and values (attentions.py:264-329):
 * q, k, v, out are (B, n_heads*dk, T); emb_k / emb_v are the shared [2*window+1][dk] tables; scores of masked
 * (query, key) pairs are set to -1e4 before the softmax as in the reference.  q, k, v share the batch stride
 * qkv_bstride (they may be row blocks of one fused projection output).  dk must be 96; T <= 1199. */
int ov_rel_attention_f32(const float* q, const float* k, const float* v, const float* emb_k, const float* emb_v,
                         const float* mask, float* out, int64_t qkv_bstride, int64_t out_bstride, int B, int n_heads,
                         int dk, int T, int ld, int window, ov_stream_t stream);

/* Depthwise dilated conv on the masked input, DDSConv.convs_sep (modules.py:102-112, :121):
 * out[b][c][t] = bias[c] + sum_j w[c][j] * (x*mask)[b][c][t + (j - (K-1)/2) * dil];  w is [C][K], K odd. */
int ov_dwconv1d_f32(const float* x, const float* w, const float* bias, const float* mask, float* out, int B, int C,
                    int T, int ld, int K, int dil, ov_stream_t stream);

/* ConvFlow.pre (1 -> C pointwise conv) + DDSConv's `x + g` (modules.py:487-488, :118-119):
 * out[b][c][t] = w[c] * x0[b][t] + bias[c] + g[b][c][t]   (g may be NULL); x0 rows are x0_bstride apart. */
int ov_expand1_f32(const float* x0, int64_t x0_bstride, const float* w, const float* bias, const float* g, float* out,
                   int B, int C, int T, int ld, ov_stream_t stream);

/* out[b][c][t] = (x[b][c][t] + bias_b[b][c]) * mask[b][t]: DurationPredictor's `x + cond(g)` and the `x * x_mask`
 * in front of its first conv (models.py:88-90). */
int ov_add_bias_mask_f32(const float* x, const float* bias_b, const float* mask, float* out, int B, int C, int T,
                         int ld, ov_stream_t stream);

/* ConvFlow.forward(reverse=True) after its projection (modules.py:493-511, transforms.py:50-188): channel c1
 * of z (B, 2, T) goes through the inverse rational-quadratic spline with linear tails whose 3*num_bins-1
 * unnormalised parameters are the first rows of h (B, >= 3*num_bins-1, T); both channels are then masked.
 * c0, c1 in {0, 1} select the physical channels (the Flips between flows are not materialised).  num_bins = 10. */
int ov_rq_spline_inverse_f32(float* z, int64_t z_bstride, int c0, int c1, const float* h, int64_t h_bstride,
                             const float* mask, int B, int T, int ld, int num_bins, int filter_channels,
                             float tail_bound, ov_stream_t stream);

/* Durations (models.py:474-479; ElementwiseAffine reverse modules.py:397-399 for the one channel that is used):
 * logw = ((z_sdp - ea_m) * exp(-ea_logs) * mask) * sdp_ratio + dp * (1 - sdp_ratio);
 * cum[b][t] = inclusive prefix sum of ceil(exp(logw) * mask * length_scale) (int32); y_len[b] = max(1, total). */
int ov_duration_f32(const float* z_sdp, int64_t z_bstride, float ea_m, float ea_logs, const float* dp,
                    int64_t dp_bstride, const float* mask, float* logw, int32_t* cum, int64_t* y_len, int B, int T,
                    int ld, float sdp_ratio, float length_scale, ov_stream_t stream);

/* generate_path + the two attn matmuls + the prior sample (models.py:480-487, commons.py:128-142): frame t' of
 * utterance b takes token j with cum[j-1] <= t' < cum[j];  z_p = m_p + noise * exp(logs_p) * noise_scale with
 * m_p = logs_p = 0 for t' >= y_len[b].  m_tok / logs_tok are (B, C, Tx) with batch stride tok_bstride (the two
 * halves of the encoder's projection).  m_p, logs_p (B, C, Ty rows ldy) and attn [B][Ty][Tx] are optional outputs. */
int ov_expand_prior_f32(const float* m_tok, const float* logs_tok, int64_t tok_bstride, int ldx, const int32_t* cum,
                        const int64_t* x_len, const int64_t* y_len, const float* noise, int64_t noise_bstride, int ldn, float* z_p,
                        float* m_p, float* logs_p, float* attn, int B, int C, int Tx, int Ty, int ldy,
                        float noise_scale, ov_stream_t stream);

/* ---- bf16 generator (BASELINE.json configs[4]; reference openvoice/models.py:272-291, modules.py:296-306) ------
 * Channels-last bf16 activations (B, L, C): C contiguous, rows dense.  bf16 values are passed as uint16_t bit
 * patterns.  Accumulation and bias are fp32; outputs are rounded to nearest even. */
typedef struct ov_conv1d_bf16_params {
  const uint16_t* x;    /* [B][L][Cin] bf16                                                              */
  const uint16_t* w;    /* packed by ov_conv1d_bf16_pack                                                  */
  const float* bias;    /* [Cout] fp32 or NULL                                                            */
  uint16_t* out;        /* [B][L][Cout] bf16                                                              */
  const uint16_t* res;  /* residual, like out, or NULL                                                    */
  const uint16_t* add;  /* second addend (MRF running sum), like out, or NULL                             */
  int32_t B, L, Cin, Cout, K, dil;   /* 'same' padding (K-1)*dil/2; Cin % 32 == 0, Cout % 32 == 0         */
  int32_t phase_s;      /* > 1: ConvTranspose as a phase conv -- Cout = phase_s * C columns ordered
                         * (phase, c); column (ph, c) of row t is written to out[b][t * phase_s + ph][c]    */
  int32_t bias_bstride; /* elements between the bias vectors of consecutive batch items (0 = shared)     */
  int32_t layout;       /* 0 = dispatcher's choice; measurement knobs: 2 = weight fragments requested one k-step
                         * ahead (round-1 scheme) instead of three; otherwise 0                     */
  float in_slope;       /* leaky-ReLU slope applied to x while staging (1.0f = identity)                  */
  float scale;          /* out = (conv + bias + res + add) * scale                                        */
  float out_slope;      /* leaky-ReLU applied to the result before rounding (0 or 1.0f = none): a tensor whose only
                         * consumer activates it (conv1 of a ResBlock pair) is stored activated, and that consumer
                         * stages it with in_slope = 1 -- no unpack / activate / repack pass in its loaders */
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][4 matrix waves][8] ticks per phase */
} ov_conv1d_bf16_params;

/* Number of bf16 elements ov_conv1d_bf16_pack writes for a dense fp32 [Cout][Cin][K] weight (0 if unsupported). */
size_t ov_conv1d_bf16_pack_size(int Cout, int Cin, int K);
/* Round a dense HOST fp32 weight to bf16 and lay it out in MFMA B-fragment order (HOST dst). */
int ov_conv1d_bf16_pack(const float* w, int Cout, int Cin, int K, uint16_t* dst);
/* Generator convs in bf16: out = (conv1d(lrelu(x)) + bias [+ res] [+ add]) * scale -- the ResBlock convs
 * (reference openvoice/modules.py:296-306, models.py:280-286; k in {3,7,11}, dilation in {1,3,5}), conv_pre (k7,
 * per-utterance bias = dec.cond, models.py:273-275) and, with phase_s, the ConvTranspose ups (models.py:278-279). */
int ov_conv1d_bf16cl(const ov_conv1d_bf16_params* p, ov_stream_t stream);
/* leaky_relu + conv_post (C -> 1, no bias) + tanh on the bf16 channels-last tensor, fp32 output [B][L]
 * (reference openvoice/models.py:287-289); w is the dense [C][K] fp32 DEVICE weight.  C = 32, K = 7. */
int ov_conv_post_tanh_bf16(const uint16_t* x, const float* w, float* out, int B, int C, int L, int K, float in_slope,
                           ov_stream_t stream);

/* One ResBlock1 iteration in ONE launch on the bf16 channels-last tensors (the fp32 twin: ov_resblock_pair_f32):
 *   out = bf16( (c2(lrelu(bf16(c1(lrelu(x)) + b1))) + b2 + x [+ add]) * scale ),  reference openvoice/modules.py:296-306.
 * The intermediate is rounded to bf16 exactly where the two ov_conv1d_bf16cl launches round it and stays in LDS;
 * results are bit-identical to that path.  w1 / w2 from ov_conv1d_bf16_pack(C, C, K), b1 / b2 fp32 [C].
 * C in {32, 64}; out must not alias x (add may alias out). */
typedef struct ov_respair_bf16_params {
  const uint16_t* x;    /* [B][L][C] bf16 */
  const uint16_t* w1;
  const float* b1;
  const uint16_t* w2;
  const float* b2;
  uint16_t* out;        /* [B][L][C] bf16 */
  const uint16_t* add;  /* MRF running sum, like out, or NULL */
  int32_t B, L, C, K, dil;
  int32_t nwg;          /* 0 = one workgroup per resident slot; n > 0 forces n workgroups (tests) */
  float slope, scale;
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][4 waves][9] ticks per phase */
} ov_respair_bf16_params;
int ov_resblock_pair_bf16cl(const ov_respair_bf16_params* p, ov_stream_t stream);
int ov_resblock_pair_bf16_supported(int C, int K, int dil);

/* One ResBlock1 iteration in ONE launch (C in {32, 64, 128}), second generation
 * (csrc/conv1d_bf16_pair2.hip).  Tensors between these launches are stored ACTIVATED:
 *   x   = bf16(lrelu(x_raw, slope))  [B][L][C]            (the producer applied the leaky ReLU before rounding)
 *   t   = bf16(lrelu(c1(x) + b1, slope))                   (never leaves the CU)
 *   out = bf16(act((c2(t) + b2 + x~) * scale)),            x~ = x >= 0 ? x : x / slope  (exact inverse in fp32),
 *         act = leaky ReLU with out_slope (1.0f: the raw sum, for a tensor that is consumed as `add` or by a kernel
 *         that activates on load);
 *   with `add` (a RAW tensor, the MRF running sum):
 *   out = bf16(act((bf16(c2(t) + b2 + x~) + add) * scale)).
 * reference openvoice/modules.py:296-306, models.py:280-286.  w1 / w2 from ov_conv1d_bf16_pack16(C, C, K), b1 / b2
 * fp32 [C].  out must not alias x; add may alias out. */
typedef struct ov_respair2_bf16_params {
  const uint16_t* x;    /* [B][L][C] bf16, activated */
  const uint16_t* w1;
  const float* b1;
  const uint16_t* w2;
  const float* b2;
  uint16_t* out;        /* [B][L][C] bf16 */
  const uint16_t* add;  /* MRF running sum (raw), like out, or NULL */
  int32_t B, L, C, K, dil;
  int32_t nwg;          /* 0 = one workgroup per CU; n > 0 forces n workgroups (tests) */
  float slope;          /* leaky-ReLU slope of x and t, 0 < slope <= 1 */
  float scale;
  float out_slope;      /* 0 or 1.0f = store the raw sum */
  int32_t exp_flags;    /* 0 in production.  MEASUREMENT ONLY: bit 0 = the loader waves idle after the first tile (wrong
                         * results); value 8 = ~500 VALU instructions of busy work per step on the loader waves */
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][4 matrix + 4 loader waves][8] ticks per phase */
} ov_respair2_bf16_params;
int ov_resblock_pair2_bf16cl(const ov_respair2_bf16_params* p, ov_stream_t stream);
/* Weights for ov_resblock_pair2_bf16cl (v_mfma_f32_16x16x32_bf16 fragment order; HOST w and dst; dst holds
 * ov_conv1d_bf16_pack_size(Cout, Cin, K) elements, the last 512 an all-zero record). */
int ov_conv1d_bf16_pack16(const float* w, int Cout, int Cin, int K, uint16_t* dst);
int ov_resblock_pair2_bf16_supported(int C, int K, int dil);

/* ---- split-precision Conv1d: fp32-level products on the bf16 matrix pipe (opt-in; csrc/conv1d_split3.h) -----------
 * An fp32 tensor is carried as THREE bf16 planes, v = hi + mid + lo with hi = bf16(v), mid = bf16(v - hi),
 * lo = bf16(v - hi - mid) (round to nearest even; lossless for fp32), channels-last and plane-major: [3][B][L][C].
 * A product x * w is the six plane products of weight >= 2^-18 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation (`products` = 6), or the three of weight >= 2^-9 on the hi / mid
 * planes only (`products` = 3: 16-bit operands).
 *   out = split3( lrelu( (conv1d(x, w) + bias [+ res~]) * scale, out_slope ) )
 * x is read as stored (the producer stores it activated); res~ = res for res_slope = 1, else the inverse leaky ReLU of
 * res (res >= 0 ? res : res / res_slope: the residual tensor is the conv input of the pair, stored activated).
 * reference: openvoice/modules.py:296-306 (xt = c1(lrelu(x)); xt = c2(lrelu(xt)); x = xt + x).
 * Cin = Cout in {64, 128, 256}, K in {3, 7, 11}, dil in {1, 3, 5} (with res: dil = 1); out must alias neither x nor res.
 * What a launch computes is bit-identical with and without col_limit. */
typedef struct ov_conv1d_split3_params {
  const uint16_t* x;     /* [3][B][L][Cin] bf16 planes */
  const uint16_t* w;     /* ov_conv1d_split3_pack(Cout, Cin, K) */
  const float* bias;     /* [Cout] fp32 */
  uint16_t* out;         /* [3][B][L][Cout] bf16 planes */
  const uint16_t* res;   /* planes like out, or NULL */
  int64_t x_plane, out_plane, res_plane;   /* elements between consecutive planes (>= B * L * C, multiples of 8) */
  int32_t B, L, Cin, Cout, K, dil;
  int32_t nwg;           /* 0 = one workgroup per CU; n > 0 forces n workgroups (tests) */
  int32_t products;      /* 6 or 3 */
  float res_slope;       /* 0 < res_slope <= 1 */
  float out_slope;       /* 0 < out_slope <= 1; 1 = store the raw result */
  float scale;
  int32_t col_limit_scale; /* with col_limit: output columns per unit of col_limit (>= 1) */
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][4 matrix waves][8] ticks per phase */
  const int32_t* col_limit; /* DEVICE int32 [B] or NULL: columns of each utterance that matter = col_limit[b] *
                             * col_limit_scale; 128-row time tiles that start at or beyond them are neither computed nor
                             * written (length-aware work lists, as ov_conv1d_params.col_limit; B <= 256, else ignored) */
} ov_conv1d_split3_params;
int ov_conv1d_split3(const ov_conv1d_split3_params* p, ov_stream_t stream);
/* Elements (uint16) of the packed three-plane weight stream; 0 when Cout or Cin is not a multiple of 32. */
size_t ov_conv1d_split3_pack_size(int Cout, int Cin, int K);
/* HOST w [Cout][Cin][K] fp32 -> 16x16x32 A-fragment order, record (((ct * Cin/32 + c) * K + tap) * 3 + plane) * 2 + f,
 * one trailing all-zero record. */
int ov_conv1d_split3_pack(const float* w, int Cout, int Cin, int K, uint16_t* dst);
int ov_conv1d_split3_supported(int Cin, int Cout, int K, int dil);
/* Layout kernels around an MRF stage that runs on ov_conv1d_split3 (the fp32 engine keeps [B][C][L], time contiguous):
 * planes [3][B][L][C] = split3(lrelu(x, slope)) of x [B][C][L] fp32 (reference: the leaky_relu of modules.py:298 moved
 * to the producer), and out [B][C][L] fp32 = (a~ [+ b~] [+ c~]) * scale, each operand de-activated with in_slope (the
 * MRF sum / mean, models.py:280-286).  C even, <= 512. */
int ov_split3_from_f32(const float* x, uint16_t* planes, int64_t plane_stride, int B, int C, int L, float slope,
                       ov_stream_t stream);
int ov_split3_to_f32(const uint16_t* a, const uint16_t* b, const uint16_t* c, int64_t plane_stride, float* out, int B,
                     int C, int L, float in_slope, float scale, ov_stream_t stream);

/* ---- Winograd-domain fp32 Conv1d (round 6; openvoice_amd/csrc/conv1d_wino.h) --------------------------------------
 * The stride-1 'same' convs of ResBlock1, reference openvoice/modules.py:296-306 -- xt = c1(lrelu(x)), xt = c2(lrelu(xt)),
 * x = xt + x -- and the MRF sum / mean of models.py:280-286, with fewer executed multiplies than the direct form:
 *   out = lrelu((conv1d(lrelu(x, in_slope), w, dilation dil) + bias [+ res] [+ add]) * scale, out_slope)
 * evaluated as ceil(K/3) shifted 3-tap groups by the minimal-filtering algorithm F(4, 3) (interpolation points 0, +-1, +-2,
 * infinity): weights transformed once in float64 at pack time, input tiles transformed in fp32 on the way into LDS, the
 * products of all groups and input channels accumulated in the transform domain on v_mfma_f32_32x32x2_f32 (six GEMMs,
 * one per point), the inverse transform + operands in the epilogue.  fp32 arithmetic throughout; executed MACs per
 * output and (co, ci): 1.5 (K = 3), 4 (K = 7), 5.75 (K = 11) instead of K (the products of a group's zero padding taps at
 * points 0 / infinity are identically zero and never issued: K = 7 is laid out with one zero tap at each end).  The transforms round where the direct conv
 * does not: against float64 the result carries ~4x the rounding error of ov_conv1d_f32 (tests/test_gpu_wino.py).
 * Tensors are fp32 [B][C][L], rows x_ld / out_ld floats apart (0 = L); L, x_ld, out_ld multiples of 4 and every
 * pointer 16-byte aligned; Cin % ov_conv1d_wino_chunk(K, Cout) == 0, Cout <= 512 and a multiple of 64 (K = 3 / 7 / 11) or
 * of 32 (K = 11); out must not alias x
 * (it may be the very tensor passed as res or add -- the MRF running sum is accumulated in place). */
typedef struct ov_conv1d_wino_params {
  const float* x;        /* [B][Cin][L] */
  const float* w;        /* ov_conv1d_wino_pack_f32(Cout, Cin, K) */
  const float* bias;     /* [Cout] natural row order; required (zeros if none) */
  float* out;            /* [B][Cout][L] */
  const float* res;      /* residual, indexed like out, or NULL */
  const float* add;      /* second addend (MRF running sum), indexed like out, or NULL */
  int64_t x_bstride, out_bstride, res_bstride, add_bstride;   /* elements between consecutive utterances */
  int32_t B, Cin, Cout, L;
  int32_t x_ld, out_ld;  /* row strides in floats; 0 = L */
  int32_t K, dil;
  int32_t nwg;           /* 0 = one workgroup per resident slot; n > 0 forces n workgroups (tests) */
  int32_t frags;         /* 128-column fragments per matrix wave: 0 = chosen by the dispatcher, else 1 or 2 (tuning knob) */
  float in_slope;        /* leaky-ReLU slope applied to x while staging (1.0f = identity) */
  float scale;
  unsigned long long* dbg; /* measurement only, NULL in production: [workgroups][8 waves][8] shader-clock ticks per phase
                          * (matrix waves: k-step loops, chunk barriers, epilogue, item set-up; helper waves: staging
                          * issue, transform, raw write, barriers); [7] = chunks */
  const int32_t* col_limit; /* DEVICE int32 [B] or NULL: length-aware work list exactly as ov_conv1d_params.col_limit --
                          * column blocks that start at or beyond col_limit[b] * col_limit_scale are neither computed
                          * nor written, what is computed is bit-identical to the full launch; B <= 256, else ignored */
  int32_t col_limit_scale;
  float out_slope;       /* leaky-ReLU slope applied to the RESULT before it is stored (the first conv of a ResBlock pair hands
                          * its consumer an activated tensor, which then stages it with in_slope = 1); 0 or 1 = none; only
                          * without res / add.  (2.09: this field was `reserved0`.) */
} ov_conv1d_wino_params;
int ov_conv1d_wino_f32(const ov_conv1d_wino_params* p, ov_stream_t stream);
/* 1 when (Cin, Cout, K, dil) has an instance, else 0 (callers then use ov_conv1d_f32). */
int ov_conv1d_wino_supported(int Cin, int Cout, int K, int dil);
/* Input channels per LDS fill of the instance for K taps and Cout rows (Cout % 128 == 0: four 32-row fragments per
 * workgroup; Cout % 64 == 0: two; Cout % 32 == 0: one, K = 11 only); the packed stream is ordered by it; 0 = no instance. */
int ov_conv1d_wino_chunk(int K, int Cout);
/* Floats of the packed transform-domain weights; 0 when the shape has no instance. */
size_t ov_conv1d_wino_pack_size(int Cout, int Cin, int K);
/* HOST w [Cout][Cin][K] fp32 -> U_p[co][g][ci] = sum_k G[p][k] w[co][ci][3g + k] (float64, rounded once to fp32) in
 * 32x32x2 A-fragment order: 1 KiB sub-record ((mt * Cin/CI + c) * (CI * G / 4) + sp) * 3 + j holds for lane l elements
 * 4j .. 4j + 3 of the 12-vector [k-step 2sp: p = 0..5][k-step 2sp + 1: p = 0..5] of row 32 mt + (l & 31) and k-row
 * 2 * kstep + (l >> 5) = g * CI + ci_local of chunk c; three zero sub-records close the stream (HOST dst). */
int ov_conv1d_wino_pack_f32(const float* w, int Cout, int Cin, int K, float* dst);

/* Library/ABI version (major*100 + minor).  2.01: ov_conv1d_params.col_limit, ov_conv_post_tanh_limited_f32,
 * ov_frame_limits_i32.  2.02: ov_unpad_rows_f32, ov_conv1d_bf16_pack16, ov_resblock_pair2_bf16cl (+ _supported).
 * 2.03: ov_conv1d_split3 (+ _pack_size, _pack, _supported), ov_split3_from_f32, ov_split3_to_f32.  2.04:
 * ov_conv1d_split3_params.col_limit / col_limit_scale.  2.05: ov_wn_layer_params.acts / row_split (the field that
 * was `reserved`; the struct grew by one pointer at its end).  2.06: ov_polyphase_fir_f32.  2.07: ov_conv1d_wino_f32 (+ _supported, _chunk,
 * _pack_size, _pack_f32).  2.08: ov_conv1d_wino_f32 instances for Cout % 32 == 0 at K = 11 (one 32-row fragment per
 * workgroup; ov_conv1d_wino_chunk(11, 32) = 2 where 2.07 returned 0).  2.09: ov_conv1d_wino_params.out_slope (the field that
 * was `reserved0`: same size and offset, 0 = none).  The Python binding
 * refuses a library older than the entry points it calls (openvoice_amd/_lib.py MIN_VERSION). */
int ov_version(void);
/* The version THIS header describes.  Parameter structs grow at their END in minor versions (2.04, 2.05, 2.07 did): a
 * caller must be compiled against the header of the library it loads -- compare ov_version() with OV_ABI_VERSION at start-up
 * and refuse a mismatch in either direction when it passes parameter structs (the Python bindings do:
 * openvoice_amd/_lib.py MIN_VERSION; the ctypes mirrors are checked field by field against this header in
 * tests/test_abi_cpu.py).  A struct is never reordered and a field never changes meaning within a major version, with
 * one exception stated here: 2.05 renamed ov_wn_layer_params.reserved to row_split AND appended `acts`, so a caller
 * built against 2.04 or older is NOT binary compatible with 2.05+ for that struct. */
#define OV_ABI_VERSION 209
/* 0 for a production build; non-zero = a measurement build with parts of the kernels compiled out (results are
 * meaningless; openvoice_amd/_lib.py refuses to load it unless OPENVOICE_AMD_ALLOW_EXPERIMENT=1). */
int ov_build_experiment(void);

#ifdef __cplusplus
}
#endif
#endif /* OPENVOICE_AMD_H */
