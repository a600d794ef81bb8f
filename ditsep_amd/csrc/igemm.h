// Implicit-GEMM descriptor shared by the engine and the MFMA kernel.
//
// One kernel family serves every dense contraction on the path:
//   Linear (DiT QKV / out-proj / FF), 1x1 conv, dilated k=7 Conv1d,
//   strided Conv1d (Oobleck encoder), ConvTranspose1d (as 2-tap phase GEMM).
// Activations are channels-last ("token-major"): row = (batch item, position),
// columns = channels, stored as P bf16 planes (common.h).  Weights are packed
// [N][taps*Cin] K-major, same planes.
//
//   out[b, j, n] = sum_{tap, ci} W[n][tap*Cin + ci] * A[b][j*in_stride + tap*tap_dil - in_pad][ci]
//
// with zero fill outside [0, Lin).  GEMM row m = b*rows_per_b + j.
// Output element offset = b*out_bstride + j*out_row_elems + out_off + n, stored
// only when 0 <= j*out_row_elems + out_off + n < out_limit (ConvTranspose1d
// phase clipping; prepend-token row shifts).
#pragma once
#include "common.h"

struct GemmDesc {
  const op16_t* A;
  long a_ps;  // plane stride of A (elements)
  const op16_t* W;
  long w_ps;  // plane stride of W
  int M, N, Cin, taps;
  int rows_per_b, Lin, in_stride, tap_dil, in_pad;
  long in_bstride;  // elements per batch item of A (normally Lin*in_row_elems)
  int in_row_elems; // elements between consecutive input rows (Cin, or more for a channel slice of a wider tensor)
  // 2-D mode (img_w > 0): rows are (y, x) of an img_h x img_w image, taps = 9 -> (dy, dx) in {-1,0,1}^2
  int img_h, img_w;
  long out_bstride;
  int out_row_elems, out_off;
  long out_limit;
  const float* bias;  // [bias_mod] or null; indexed n % bias_mod (packed order when swiglu)
  int bias_mod;
  const float* resid;  // fp32 or null; addressed b*resid_bstride + j*resid_row_elems + resid_off + n
  long resid_bstride;
  int resid_row_elems, resid_off;
  const float* bbias;  // per-(batch item, channel) bias [B][bbias_stride] or null (time-embedding bias)
  int bbias_stride;
  // GroupNorm statistics of the fp32 output, written by the epilogue WITHOUT atomics: gn_stats[B][S][N/4][2] holds,
  // per item, per 64-row slice s of the item (S = rows_per_b / 64) and per 4-channel quad, (mean, sum of squared
  // deviations from that mean) of the 256 values -- exact two-pass values from the accumulator registers.  The
  // consumer (gn_apply) combines the slices x quads of a group in a fixed order (Chan's parallel formula):
  // bit-reproducible, no E[x^2] - E[x]^2 cancellation.  Needs rows_per_b % 64 == 0, N % 4 == 0.  null = off
  float* gn_stats;
  // the same partials a second time, in the layout of a CONCAT buffer this output is a channel slice of
  // ([B][S][gn_nq2][2], this tensor's quads starting at gn_qoff2): the GroupNorm over the concatenation then needs no
  // statistics pass of its own.  null = off
  float* gn_stats2;
  int gn_nq2, gn_qoff2;
  float* out_f32;      // or null
  // (out_scale sits in the 8-byte-aligned half of its slot on purpose: with f32_op first, kernels that compile f32_op out
  // fetched the kernarg chunk [out_scale | out_planes | ...] as one 16-byte load starting at an odd dword, could not
  // hold the pointer in an aligned SGPR pair, parked the chunk in SCRATCH and re-read it -- with s_waitcnt vmcnt(0), i.e.
  // behind every store already issued -- once per 16x16 output tile of the epilogue)
  float out_scale;     // applied after bias + residual
  int f32_op;          // DSN_F32_*
  op16_t* out_planes;  // or null: act(out) written as P planes
  long out_ps;
  int act;             // DSN_ACT_*
  const float* act_a;  // snake alpha   (exp applied) [act_mod]
  const float* act_b;  // snake 1/(beta+1e-9)         [act_mod]
  int act_mod;
  int swiglu;          // packed N holds (value16, gate16) interleaved groups; N_out = N/2
  int tiles_m, tiles_n;
  // split-K: the k-tiles are divided over `ksplit` workgroups per output tile; slice z writes
  // its raw fp32 partial sums (no bias / residual / activation) to out_f32 + z*slab_stride.
  // The consumer (residual_norm kernel) adds bias + residual + slabs.
  int ksplit;
  long slab_stride;
  // fused QKV epilogue (DiT): rotary tables [rope_S][32] (cos | sin), token position = m % rope_S,
  // sections q | k | v of width qkv_D, 64-wide heads; q scaled by q_scale.  null = off.
  const float* rope_cos;
  const float* rope_sin;
  int rope_S, qkv_D;
  float q_scale;
  // fp8 (MX) operands, row-panel kernel only: A and W hold OCP e4m3 bytes (addressed as op16_t with every element
  // count -- Cin, in_row_elems, strides -- given in PAIRS of bytes, i.e. K/2), a_scale [M][mx_kblocks] and
  // w_scale [N][mx_kblocks] one E8M0 byte per 32 K-elements.  null = 16-bit operands.
  const unsigned char* a_scale;
  const unsigned char* w_scale;
  int mx_kblocks;
  // fp8 output of the SwiGLU epilogue (the next GEMM's A operand): e4m3 bytes [M][N/2] + E8M0 [M][N/64]
  unsigned char* out_fp8;
  unsigned char* out_fp8_scale;
  // LayerNorm folded into the consuming GEMM (row-panel kernel, single-plane modes).
  // Producer side (a GEMM whose fp32 output is the residual stream x'): stat_out [M][stat_np][2] receives, per row and
  // per 64-column wave slice, (mean, sum of squared deviations from that mean) of x' -- exact two-pass values from the
  // accumulator registers; N % 64 == 0, stat_np = N / 64.
  float* stat_out;
  int stat_np;
  // Consumer side: A holds the RAW x' operand plane, W = W_orig * diag(gamma); the epilogue applies
  //   LN(x') W_orig^T = rstd * (x' W^T - mean * colsum) (+ bias, which already carries beta W_orig^T)
  // with (mean, rstd) per row combined from ln_stats [M][ln_np][2] (Chan's parallel formula, fixed order) into LDS by
  // the kernel prologue, and colsum[n] = sum_k W[n][k] of the ROUNDED packed weights.  null = off.
  const float* ln_stats;
  int ln_np;
  const float* ln_colsum;
  float ln_eps;
  int panel_rows;  // row-panel kernel only: rows per workgroup (<= 272)
  int panel_wm;    // row-panel kernel: wave rows (0 / 4: 4 x WN waves; 2: the 8-wave variants, 256-column tiles)
  // 1x1 shortcut conv accumulated into the same tile before the 3x3 taps (igemm_halo3x3_kernel, sc_A != null): the
  // NIN shortcut of a ResnetBlockBigGANpp, out = (Conv_1(a1) + Conv_2(x)) / sqrt 2, without the fp32 round trip of
  // Conv_2's output and without its launch.  sc_A: planes [B][rows_per_b][sc_row_elems] (first sc_Cin channels used),
  // sc_W: packed [N][sc_Cin], sc_bias [N] (added with the other biases; null: none).
  const op16_t* sc_A;
  const op16_t* sc_W;
  const float* sc_bias;
  int sc_Cin, sc_row_elems;
  long sc_bstride;
  // GroupNorm finished by the producing conv (igemm_halo3x3_kernel, gnf_out != null; needs gn_stats): the row-tile
  // workgroups of an image publish their slice partials write-through, meet at the (image, column tile) counter, combine
  // the image's partials exactly as gn_apply_kernel does, and store silu(GroupNorm(out)) as the operand plane of the next
  // conv straight from their accumulators -- the fp32 tensor is never written and the gn_apply launch is not needed.
  // All workgroups of the grid must be resident together (igemm_halo3x3_gnfin_ok).
  op16_t* gnf_out;           // [B * rows_per_b][N] 16-bit plane
  const float* gnf_gamma;    // [N]
  const float* gnf_beta;     // [N]
  unsigned* gnf_sync;        // [B * tiles_n] counters, only ever incremented (zeroed once at allocation)
  int* gnf_err;              // host-visible flag, set when a wait gave up
  float gnf_eps;
  int gnf_silu;
  int cfg_bm, cfg_bn, cfg_nst, cfg_bk;  // explicit tile configuration for igemm2_launch (0 = heuristic)
  int dbg;         // development: 1 = skip in-loop glds (compute only), 2 = skip MFMAs (staging only)
  int m_fast;      // tile order inside an XCD's share: 1 = row panels fastest (few rows, many columns)
};

// launchers (igemm.hip)
hipError_t igemm_launch(const GemmDesc& d, int pl, hipStream_t stream);   // v1: register-staged
hipError_t igemm2_launch(const GemmDesc& d, int pl, hipStream_t stream);  // v2: glds ring + split-K, auto tile
hipError_t igemm2_launch_cfg(const GemmDesc& d, int pl, int bm, int bn, int nstage, int bk, hipStream_t stream);
// sums `nslab` split-K slabs (written by a GEMM launched with d.ksplit = nslab, d.out_f32 = slabs) and runs the ordinary
// epilogue of `d` on the result (single-plane modes, N % 64 == 0)
hipError_t igemm_slab_epilogue_launch(const GemmDesc& d, int pl, const float* slabs, int nslab, long slab_stride,
                                      hipStream_t stream);
// halo-resident 3x3 conv (single-plane modes, W <= 32, H*W % 256 == 0); hipErrorNotSupported when not eligible
hipError_t igemm_halo3x3_launch(const GemmDesc& d, int pl, hipStream_t stream);
// can `d` (with gnf_out set) run as a producer-finished GroupNorm conv: halo-kernel eligible and one resident round?
bool igemm_halo3x3_gnfin_ok(const GemmDesc& d, int pl);
bool igemm2_gnfin_ok(const GemmDesc& d, int pl);   // the same inside igemm2's 128 x 64 tile (128-pixel images, no hand-off)
// will igemm2_launch run `d` on the halo kernel (the only one that takes sc_A)?
bool igemm_halo3x3_eligible(const GemmDesc& d, int pl);
// `pl` = DSN_PL(plane count, fp16 flag)
// row-panel variant (igemm.hip): d.panel_rows rows x bn (128 | 256) columns per workgroup
hipError_t igemm_panel_launch(const GemmDesc& d, int pl, int bn, hipStream_t stream);
// skinny variant for M <= 48 rows (single-plane modes, plain row-major GEMM): one wave per 32 columns x split-K,
// weights streamed straight into MFMA fragments
hipError_t igemm_skinny_launch(const GemmDesc& d, int pl, hipStream_t stream);
// the same with fp8 (MX) operands: d.a_scale / d.w_scale set, d.Cin = K/2 (byte pairs), K % 128 == 0
hipError_t igemm_panel_fp8_launch(const GemmDesc& d, int bn, hipStream_t stream);

// DiT to_qkv GEMM + rotary embedding + self-attention in one launch (qkv_attn.hip): row panels of `ipp` whole items
// (ipp * S <= qkv_attention_max_rows()) x one 64-wide head per workgroup.  Single-plane 16-bit modes.
struct QkvAttnDesc {
  const op16_t* A;        // LayerNorm output operand plane [M][D]
  const op16_t* W;        // to_qkv packed [3 D][D] K-major: q | k | v sections, head-major inside a section
  const float* bias;      // [3 D] or null
  const float* rope_cos;  // [S][32]
  const float* rope_sin;
  op16_t* out;            // attention output operand plane [M][D] (must not alias A), or null with out8
  unsigned char* out8;    // fp8 (MX) output instead: e4m3 bytes [M][D] + E8M0 block scales [M][D/32] (fp8 out-projection)
  unsigned char* out8_scale;
  int M, D, H, S;         // token rows (items x S), model width = 64 H, heads, tokens per item
  int ipp;                // items per panel
  int panels;             // set by the launcher
  float q_scale;          // 1 / sqrt(64)
};
hipError_t qkv_attention_launch(const QkvAttnDesc& d, int pl, hipStream_t stream);
int qkv_attention_max_rows();            // tallest panel (240)
int qkv_attention_panel_rows(int rows);   // tile height the launcher picks for panels of `rows` rows (144 | 240)

// Fused Oobleck ResidualUnit over 128-channel channels-last sequences (ru_fused.hip):
//   out = X + conv1x1(act_mid(conv_k7_dil(A) + b7)) + b1 ;  planes(out) carry act_out for the consumer.
// A / out_planes must not alias (neighbouring workgroups read each other's halo rows); X / out_f32 may.
struct RuDesc {
  const op16_t* A;  // act(x) operand planes [S][L][128]
  long a_ps;
  const float* X;  // fp32 residual stream [S][L][128]
  const op16_t* W7;  // [128][7*128] K-major planes
  long w7_ps;
  const float* b7;
  const op16_t* W1;  // [128][128]
  long w1_ps;
  const float* b1;
  float* out_f32;      // may be null
  op16_t* out_planes;  // may be null
  long out_ps;
  int act_mid, act_out;  // DSN_ACT_*
  const float *mid_a, *mid_b, *out_a, *out_b;  // snake alpha / 1/beta per channel
  int S, L, dil;
};
hipError_t ru_fused_launch(const RuDesc& d, int pl, hipStream_t stream);
