// Launch wrappers for the non-GEMM kernels of the separation path (kernels.hip).
#pragma once
#include "common.h"

// ---- layout transforms -----------------------------------------------------
// src0 [B][C0][T] (+ src1 [B][C1][T]) channel-major fp32 -> token-major
// dst[(b*T + t)*(C0+C1) + c] as fp32 (optional) and as operand planes (optional).
void launch_pack_tokens(const float* src0, int C0, const float* src1, int C1, int B, int T,
                        float* dst_f32, op16_t* dst_planes, long ps, int planes, hipStream_t s);
// rows of `width` floats (multiple of 4) into rows `dst_stride` floats apart
void launch_copy_rows(const float* src, float* dst, int rows, int width, long dst_stride, hipStream_t s);
// token-major fp32 [B*T][C] -> channel-major [B][C][T]
void launch_unpack_tokens(const float* src, float* dst, int B, int C, int T, hipStream_t s);
// fp32 [n] -> operand planes, optional activation
void launch_to_planes(const float* src, op16_t* dst, long ps, int planes, long n, hipStream_t s);

// ---- predictor-corrector sampler (OUVE; reference layout x[B,n,D,T], y[B,1,D,T]) ----
// score is token-major [B*T][n*D].
// mean: y [B,1,D,T] (mean_full = 0) or a full prior mean [B,n,D,T] (mean_full = 1)
void launch_pc_prior(const float* mean, int mean_full, const float* z, float* x, float stdT, int B, int n, int D,
                     int T, hipStream_t s);
// norms == null: host scalars (ALD); else [2][B] per-item norms of score and noise (Langevin corrector)
void launch_pc_corrector(float* x, float* x_mean /*nullable*/, const float* score_tok, const float* z, float step,
                         float noise_gain, const float* norms, float snr, int B, int n, int D, int T, hipStream_t s);
void launch_pc_item_norms(const float* a, long per_item, int B, float* out, hipStream_t s);
// em = 0 reverse diffusion, 1 Euler-Maruyama; G = g sqrt(dt)
void launch_pc_predictor(float* x, float* x_mean, const float* y, const float* score_tok, const float* z,
                         float theta, float dt, float G, float g, int em, int B, int n, int D, int T, hipStream_t s);

// ---- secondary sampler family (MixSDE / PriorMixSDE + ald2, Schroedinger bridge) on x [B,n,D,T] ----
// smix [B][D*T] = PriorMixSDE._std_sigma_mix of the flattened mixture latent (null: MixSDE, factor 1)
void launch_sigma_mix(const float* y, float* smix, int B, int L, int avg_len, hipStream_t s);
// x = 0.5 y + (s1 A + s2 Pn) z * smix
void launch_mix_prior(const float* y, const float* z, float* x, const float* smix, float s1, float s2, int B, int n, int D,
                      int T, hipStream_t s);
// ald2 step with L = (sq1 A + sq2 Pn) smix (x_mean may be null)
void launch_mix_corrector(float* x, float* x_mean, const float* score_tok, const float* z, const float* smix, float sq1,
                          float sq2, float snr, int B, int n, int D, int T, hipStream_t s);
// em = 0 reverse diffusion, 1 Euler-Maruyama; g = diffusion scalar, sqdt = sqrt(dt)
void launch_mix_predictor(float* x, float* x_mean, const float* score_tok, const float* z, const float* smix,
                          float lambda, float dt, float g, float sqdt, int em, int B, int n, int D, int T, hipStream_t s);
// x = w_prev x + w_est est + w3 (third_is_y ? y[B,1,D,T] : z[B,n,D,T]); third may be null
void launch_sb_update(float* x, const float* est_tok, const float* third, float w_prev, float w_est, float w3,
                      int third_is_y, int B, int n, int D, int T, hipStream_t s);
void launch_repeat_sources(const float* y, float* x, int B, int n, int D, int T, hipStream_t s);

// ---- DiT pieces ---------------------------------------------------------------
// x += bias + sum(split-K slabs) (written back when nslab > 0), then LayerNorm (do_norm) or a
// plain copy to operand planes.  D <= 4096, D % 4 == 0.
// out_fp8_scale != null: `out` receives fp8 (e4m3) bytes [rows][D] and out_fp8_scale E8M0 scales [rows][D/32].
void launch_residual_norm(float* x, const float* slabs, int nslab, long slab_stride, const float* bias,
                          const float* gamma, const float* beta, op16_t* out, long ps, int planes, int rows, int D,
                          float eps, int do_norm, hipStream_t s, unsigned char* out_fp8_scale = nullptr);
// FourierFeatures: t[B], w[half] -> planes [B][2*half] = [cos(2 pi t w), sin(2 pi t w)]
void launch_timestep_features(const float* t, const float* w, int B, int half, op16_t* out, long ps,
                              int planes, hipStream_t s);
// MFMA attention over operand planes q|k|v [B*S][3*H*64] written by the fused QKV epilogue
// (attention.hip); S <= 256.
// out_fp8_scale != null: `out` receives fp8 (e4m3) bytes [B*S][H*dh] and out_fp8_scale the E8M0 block scales
// [B*S][H*dh/32] (MX operand of the fp8 out-projection) instead of 16-bit planes.
int launch_attention_mfma(const op16_t* qkv, long ps, op16_t* out, long out_ps, int pl, int B, int S, int H, int dh,
                          hipStream_t s, unsigned char* out_fp8_scale = nullptr);   // dh in {64,128,256}; -1: unsupported width
void launch_rope_tables(float* cos_t, float* sin_t, int S, int rot, hipStream_t s);

// ---- Oobleck edges -------------------------------------------------------------
// final decoder conv: planes [S*L][C] (already activated), w fp32 [7][C] -> tanh(sum) fp32 [S*L]
void launch_conv_out1(const op16_t* a, long ps, int planes, const float* w, float* out, int S, int L, int C,
                      int ktaps, int apply_tanh, hipStream_t s);
// first encoder conv: wav fp32 [S][L] (Cin = 1), w [Cout][K], bias -> fp32 x [S*L][Cout] + act planes
void launch_conv_in1(const float* wav, const float* w, const float* bias, int S, int L, int Cout, int ktaps,
                     float* out_f32, op16_t* out_planes, long ps, int planes, int act, const float* act_a,
                     const float* act_b, hipStream_t s);
// VAE bottleneck sample: enc fp32 token-major [S*T][2*D] (mean | scale) + noise [S][D][T] -> y [S][D][T]
void launch_vae_sample(const float* enc_tok, const float* noise, float* y, int S, int D, int T, hipStream_t s);

// ---- RNG ------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller standard normals
void launch_randn(float* out, long n, unsigned long long seed, unsigned long long offset, hipStream_t s);

// ---- weight packing ---------------------------------------------------------------
enum { PACK_LINEAR = 0, PACK_LINEAR_SWIGLU = 1, PACK_CONV = 2, PACK_CONVT = 3, PACK_CONV2D = 4, PACK_NIN = 5 };
// PACK_CONV2D: src [Cout][Cin][taps] (taps = kw: 9 for 3x3 row-major (ky,kx), 1 for 1x1) -> dst [N][taps*Cin_pad],
//   Cin_pad = `stride`, zero where ci >= Cin or n >= Cout.   PACK_NIN: src [Cin][Cout] -> dst [Cout][Cin].
// per-row scale g / ||v|| for old-style weight norm (norm over all dims but 0); v [R][inner]
void launch_wn_scale(const float* v, const float* g, float* scale, int R, long inner, hipStream_t s);
// generic gather into packed [N][K] planes; see kernels.hip for the index maps
// colscale (PACK_LINEAR / PACK_LINEAR_SWIGLU only): per input column factor, W[n][k] * colscale[k] (LayerNorm gamma
// folded into the consuming Linear)
void launch_pack_weight(const float* src, const float* scale, op16_t* dst, long ps, int planes, int mode,
                        int N, int K, int Cin, int Cout, int kw, int stride, hipStream_t s,
                        const float* colscale = nullptr);
// out[n] = sum_k packed[n][k] (the rounded operand values; both planes in the split modes)
void launch_packed_row_sum(const op16_t* w, long ps, int planes, int N, int K, float* out, hipStream_t s);
// out[n] = (bias ? bias[n] : 0) + sum_k W[n][k] beta[k]
void launch_bias_plus_wbeta(const float* W, const float* beta, const float* bias, int N, int K, float* out,
                            hipStream_t s);
void launch_pack_bias_swiglu(const float* src, float* dst, int N, hipStream_t s);
// Linear weight [N][K] fp32 (swiglu: rows interleaved as PACK_LINEAR_SWIGLU) -> fp8 e4m3 bytes [N][K] + E8M0 block
// scales [N][K/32] (one per 32 consecutive K-elements); K % 32 == 0.  colscale [K] (optional) is multiplied in before
// quantisation (a LayerNorm gamma folded into the weight).
void launch_pack_weight_fp8(const float* src, unsigned char* dst, unsigned char* scales, int N, int K, int swiglu,
                            hipStream_t s, const float* colscale = nullptr);
// out[n] = sum_k of the dequantised fp8 (MX) row n: the folded LayerNorm's mean * colsum term for fp8 weights
void launch_fp8_row_sum(const unsigned char* w, const unsigned char* scales, int N, int K, float* out, hipStream_t s);
// snake parameters: alpha -> exp(alpha), beta -> 1/(exp(beta)+1e-9)
void launch_snake_params(const float* alpha, const float* beta, float* a_out, float* ib_out, int C,
                         hipStream_t s);

// ---- NCSN++ (ncsn_kernels.hip): channels-last images, row = (item, y, x) -------------------------
// Views: x + item*bstride + row*rstride + channel (rstride >= C lets a kernel read a channel slice
// of a wider concat buffer).
void launch_ncsn_pack(const float* xt, const float* mix, int B, int n, int H, int T, int Wp, int Cp, float* of,
                      op16_t* op, long ps, int planes, hipStream_t s);
void launch_gn_stats(const float* x, long bstride, int rstride, int C, int G, int B, int HW, float* stats,
                     hipStream_t s);   // stats [B][ceil(HW/64)][C/4][2] = (mean, M2) per 64-row slice and channel quad
void launch_gn_apply(const float* x, long bstride, int rstride, int C, int G, int B, int HW, const float* stats,
                     const float* gamma, const float* beta, float eps, int silu, float* of, op16_t* op, long ps,
                     int planes, hipStream_t s);
void launch_fir2d(const float* x, long bstride, int rstride, int C, int B, int H, int W, int up, const float* add,
                  float* of, op16_t* op, long ps, int planes, hipStream_t s);
void launch_ncsn_fourier(const float* t, const float* w, int B, int nf, op16_t* out, long ps, int planes,
                         hipStream_t s);
void launch_ncsn_output(const float* pyr, int Cp, const float* t, const float* w, const float* bias, int cin, int n,
                        int B, int H, int T, int Wp, float* score, hipStream_t s);

// ---- metrics -----------------------------------------------------------------------------------------
// out[(b*n + i)*n + j][3] = (<ref_i, est_j>, |ref_i|^2, |est_j|^2) in fp64
void launch_sisdr_dots(const float* ref, const float* est, int B, int n, int L, double* out, hipStream_t s);
