/* C-ABI of libditsep_hip.so -- MI355X (gfx950) latent-diffusion separation engine.
 *
 * Drop-in boundary for the hot path of eduardburlacu/DiTSep:
 *   LatentDiffSep.separate()            reference src/diffsep_latent.py:471-487
 *   LatentDiffSep.get_pc_sampler()      reference src/diffsep_latent.py:406-469
 *     -> sdes.get_pc_sampler()          reference src/sdes/__init__.py:133-193
 *   LatentDiffSep.forward (score net)   reference src/diffsep_latent.py:147-148
 *   LatentDiffSep.encode / decode       reference src/diffsep_latent.py:107-128
 *
 * Conventions: plain pointers and sizes only; every data pointer is a DEVICE
 * pointer owned by the caller (fp32, contiguous, the reference's own tensor
 * layouts) unless a parameter says host; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Functions return 0 on success, a negative
 * DSN_E* code on failure (message via dsn_last_error); no exceptions cross the
 * boundary.  One context per device; a context is not re-entrant.
 */
#ifndef DITSEP_HIP_H
#define DITSEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSN_OK 0
#define DSN_EINVAL (-1)   /* bad argument / shape                                  */
#define DSN_EHIP (-2)     /* HIP runtime error                                     */
#define DSN_ESTATE (-3)   /* weights missing / not finalized                       */
#define DSN_ENOMEM (-4)

#define DSN_PREC_BF16 1    /* bf16 MFMA operands, fp32 accumulate                  */
#define DSN_PREC_BF16X3 2  /* split-bf16 (hi,lo) operands: 3 bf16 MFMAs per product */
#define DSN_PREC_FP16 3    /* fp16 MFMA operands (11-bit mantissa), fp32 accumulate */
#define DSN_PREC_FP16X3 4  /* split-fp16 (hi,lo) operands: 3 fp16 MFMAs per product */
#define DSN_PREC_FP8 5     /* BASELINE config 5: the DiT layer GEMMs (to_qkv, to_out, FF in/out) on fp8 e4m3 operands
                              with E8M0 block scales per 32 K-elements (v_mfma_scale_f32_16x16x128_f8f6f4, fp32
                              accumulate); everything else as DSN_PREC_FP16.  A throughput mode: e4m3 operand rounding
                              (2^-4) cannot meet the 1e-3 waveform bound, its deviation is reported, not hidden. */

#define DSN_SCORE_NONE 0
#define DSN_SCORE_DIT 1    /* reference src/stable_audio_tools/models/dit.py:12-244  */
#define DSN_SCORE_NCSNPP 2 /* reference src/models/diffsep/score_models.py:140-186   */

#define DSN_MAX_VAE_BLOCKS 8

typedef struct dsn_ctx dsn_ctx;

typedef struct dsn_config {
  int32_t device;
  int32_t precision;   /* DSN_PREC_*                                               */
  int32_t n_src;       /* config.model.n_speakers                                  */
  int32_t latent_dim;  /* VAE latent channels (64)                                 */
  /* score network */
  int32_t score_kind;  /* DSN_SCORE_*                                              */
  int32_t dit_embed_dim, dit_depth, dit_heads;
  /* NCSN++ latent score network (config.model.score_model.backbone_args, default.yaml:16-28) */
  int32_t ncsn_nf, ncsn_n_levels;
  int32_t ncsn_ch_mult[4];
  int32_t ncsn_num_res_blocks, ncsn_attn_resolution, ncsn_image_size, ncsn_max_latent_length;
  /* Oobleck VAE (oobleck_finetune.json keys) */
  int32_t vae_channels;
  int32_t vae_n_blocks;                       /* len(c_mults) == len(strides)      */
  int32_t vae_c_mults[DSN_MAX_VAE_BLOCKS];
  int32_t vae_strides[DSN_MAX_VAE_BLOCKS];
  int32_t vae_enc_latent_dim;                 /* encoder out channels (2*latent)   */
  int32_t vae_use_snake, vae_final_tanh;
  int32_t vae_has_encoder, vae_has_decoder;
  /* OUVE SDE (config.model.sde) */
  float sde_theta, sde_sigma_min, sde_sigma_max;
} dsn_config;

/* lifecycle */
dsn_ctx* dsn_create(const dsn_config* cfg);
void dsn_destroy(dsn_ctx* ctx);
const char* dsn_last_error(const dsn_ctx* ctx);   /* ctx may be NULL after a failed dsn_create */

/* Weights: one call per state_dict entry, reference key names and layouts
 * (Lightning checkpoint `state_dict`: `score_model.*`, `vae.decoder.*`,
 * `vae.encoder.*`; old-style weight norm `weight_g` / `weight_v`;
 * reference src/diffsep_latent.py:341-392).  `data` is fp32; is_device != 0 when it
 * is a device pointer.  dsn_finalize_weights folds weight norm, transposes to the
 * K-major packed MFMA layout and splits into bf16 planes, on the GPU. */
int dsn_load_tensor(dsn_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim,
                    int is_device);
int dsn_finalize_weights(dsn_ctx* ctx);
/* strict != 0 (what dsn_finalize_weights does): besides a missing tensor, any loaded tensor the configured network
 * does not consume is an error (DSN_ESTATE, the names in dsn_last_error) -- nn.Module.load_state_dict(strict=True)
 * semantics; known module buffers (`*.inv_freq`, `*.num_batches_tracked`) are ignored.  strict == 0 drops them. */
int dsn_finalize_weights_ex(dsn_ctx* ctx, int strict);

/* score = score_model(xt, time_cond, mix)      (LatentDiffSep.forward)
 * xt [B,n_src,D,T], t [B], mix [B,1,D,T] -> out [B,n_src,D,T] */
int dsn_score(dsn_ctx* ctx, const float* xt, const float* t, const float* mix, float* out, int B, int T,
              void* stream);

/* OUVE schedule scalars exactly as the sampler uses them (host arrays of N floats each;
 * any output pointer may be NULL).  timesteps = linspace(1, t_eps, N). */
int dsn_ouve_schedule(const dsn_ctx* ctx, int N, float t_eps, float snr, float* timesteps, float* std,
                      float* corr_step, float* corr_gain, float* G, float* std_T);

/* pc_sampler() of sdes.get_pc_sampler("reverse_diffusion", "ald", ...):
 * y [B,1,D,T] -> x_out [B,n_src,D,T]; nfe_out (host) = N*(corrector_steps+1).
 * noise: [1 + N*(corrector_steps+1), B, n_src, D, T] standard normals in the reference's
 * draw order (prior, then per step corrector draws, predictor draw), or NULL to draw
 * on-device (Philox4x32-10, `seed`). */
int dsn_pc_sample(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T,
                  int N, int corrector_steps, float snr, float t_eps, int denoise, int* nfe_out, void* stream);

/* pc_sampler() of sdes.get_pc_scheduled_sampler (src/sdes/__init__.py:49-130): as dsn_pc_sample with
 * caller-provided timesteps (host array, N entries used: the first N of the reference's N+1-point
 * linear / log / revlog grid).  State shape is [B,n_src,D,T] (the reference samples y.shape there). */
int dsn_pc_sample_sched(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T,
                        int N, const float* timesteps_host, int corrector_steps, float snr, int denoise, int* nfe_out,
                        void* stream);

/* The other registered predictors / correctors of the reference's sampler library, same loop:
 *   predictor  DSN_PRED_REVERSE_DIFFUSION (predictors.py:55-66) | DSN_PRED_EULER_MARUYAMA (:39-52) |
 *              DSN_PRED_NONE (:69-77, no score call, no draw)
 *   corrector  DSN_CORR_ALD (correctors.py:58-84) | DSN_CORR_LANGEVIN (:35-55: one step size per corrector step from
 *              the batch means of the per-item score / noise norms -- it couples the items of a batch)
 * noise (or the on-device draw) holds 1 + N*(corrector_steps + (predictor != NONE)) tensors in consumption order.
 * timesteps: host array of N floats or NULL (= linspace(1, t_eps, N)).  prior_mean: optional [B,n_src,D,T] device
 * tensor the prior is drawn around instead of y (`true_mean`, sdes/__init__.py:175-176).  intermediates: optional
 * [N][2][B,n_src,D,T] device buffer receiving (x, x_mean) after each step's corrector (`intermediate=True`).
 * nfe_out = N*(corrector_steps+1) whatever the predictor, as the reference reports it (__init__.py:186). */
enum { DSN_PRED_REVERSE_DIFFUSION = 0, DSN_PRED_EULER_MARUYAMA = 1, DSN_PRED_NONE = 2 };
enum { DSN_CORR_ALD = 0, DSN_CORR_LANGEVIN = 1 };
typedef struct dsn_sampler_opts {
  int predictor, corrector, corrector_steps;
  float snr, t_eps;
  int denoise;
  const float* timesteps;
  const float* prior_mean;
  float* intermediates;
} dsn_sampler_opts;
int dsn_pc_sample_ex(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T,
                     int N, const dsn_sampler_opts* opts, int* nfe_out, void* stream);

/* The secondary SDE family of the reference's sampler package on the latent state read as [B, n_src, D*T]:
 * MixSDE (sdes.py:182-352) / PriorMixSDE (:355-593; diffusion scaled by the running RMS of the mixture over `avg_len`
 * flattened latent samples) through the same predictor-corrector loop, predictors as above, corrector
 * DSN_MIXCORR_ALD2 (AnnealedLangevinDynamics2, correctors.py:87-121) or DSN_MIXCORR_NONE.  noise: [1 + N*(corrector_steps
 * + (predictor != NONE)), B, n_src, D, T] in consumption order, or NULL (device RNG).  MixSDE's prior is written for
 * 2 sources (reference :347).  nfe_out = N*(corrector_steps+1). */
enum { DSN_MIXCORR_ALD2 = 0, DSN_MIXCORR_NONE = 1 };
typedef struct dsn_mix_opts {
  int prior_mix;                      /* 0 MixSDE, 1 PriorMixSDE */
  float d_lambda, sigma_min, sigma_max;
  int avg_len;                        /* PriorMixSDE only */
  int predictor, corrector, corrector_steps;
  float snr, t_eps;
  int denoise;
} dsn_mix_opts;
int dsn_pc_sample_mix(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T, int N,
                      const dsn_mix_opts* opts, int* nfe_out, void* stream);

/* get_sb_sampler (src/sdes/__init__.py:284-389) with SBVESDE(k, c, eps = sb_eps) (sdes.py:701-779): the state starts
 * as y repeated over the sources; N first-order Schroedinger-bridge steps over linspace(1, t_eps, N + 1), the score
 * network's output taken as the data estimate.  sampler_type DSN_SB_SDE consumes noise [N, B, n_src, D, T] (or the
 * device RNG), DSN_SB_ODE none. */
enum { DSN_SB_SDE = 0, DSN_SB_ODE = 1 };
int dsn_sb_sample(dsn_ctx* ctx, const float* y, const float* noise, uint64_t seed, float* x_out, int B, int T, int N, float k,
                  float c, float sb_eps, float t_eps, int sampler_type, void* stream);

/* LatentDiffSep.decode: est [B,n_src,D,T] -> wav [B,n_src,target_len] (crop of hop*T;
 * target_len <= 0 means hop*T). */
int dsn_decode(dsn_ctx* ctx, const float* est, float* wav, int B, int T, int target_len, void* stream);

/* AudioAutoencoder.decode_audio(latents, chunked=True, overlap, chunk_size) (reference
 * src/stable_audio_tools/models/autoencoders.py:665-731): long-form decode as independent chunks of `chunk_size`
 * latent frames every chunk_size-overlap frames (plus a final chunk flush with the end), pasted with overlap/2
 * frames dropped at each interior edge.  Bounds activation memory by the chunk, not the utterance.
 * T < chunk_size is an error (the reference fails there too). */
int dsn_decode_chunked(dsn_ctx* ctx, const float* est, float* wav, int B, int T, int target_len, int chunk_size,
                       int overlap, void* stream);

/* AudioAutoencoder.encode_audio(audio, chunked=True, ...) (autoencoders.py:596-663) behind LatentDiffSep.encode's
 * padding: the encoder output (mean ++ scale) is stitched by the same rule, then sampled once with vae_noise
 * [B,D,T] (the reference samples every chunk with its own draw before stitching: the same distribution). */
int dsn_encode_chunked(dsn_ctx* ctx, const float* mix, const float* vae_noise, uint64_t seed, float* y, int B, int L,
                       int chunk_size, int overlap, void* stream);

/* LatentDiffSep.encode (mixture branch): mix [B,1,L] -> y [B,1,D,T], T = (L + pad)/hop with
 * the reference's pad rule (a full extra hop when L % hop == 0).  vae_noise [B,D,T] or NULL
 * (on-device draw with `seed`). */
int dsn_encode(dsn_ctx* ctx, const float* mix, const float* vae_noise, uint64_t seed, float* y, int B, int L,
               void* stream);
int dsn_latent_frames(const dsn_ctx* ctx, int L);   /* T for an L-sample mixture */
int dsn_hop_length(const dsn_ctx* ctx);

/* LatentDiffSep.separate(mix, target_dim, latent=False): encode -> sample -> decode.
 * noise / vae_noise as above (both NULL = on-device RNG). */
int dsn_separate(dsn_ctx* ctx, const float* mix, const float* vae_noise, const float* noise, uint64_t seed,
                 float* wav, int B, int L, int target_len, int N, int corrector_steps, float snr, float t_eps,
                 int denoise, int* nfe_out, void* stream);

/* SI-SDR with permutation-invariant assignment: ref, est [B,n,L] (device) -> si_sdr [B,n] and perm [B,n]
 * (host; est source perm[b][i] is matched to ref source i).  Replaces the fast_bss_eval call of
 * src/evaluate_latent.py:118-136 (compute_permutation=True, zero_mean=False); n <= 4. */
int dsn_si_sdr_pit(dsn_ctx* ctx, const float* ref, const float* est, int B, int n, int L, float* si_sdr_out,
                   int* perm_out, void* stream);

/* SI-SDR, SI-SIR and SI-SAR with the permutation solved: what evaluate_latent.py:118-136 gets from
 * fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True, clamp_db=100).
 * ref, est [B,n,L] (device) -> si_sdr / si_sir / si_sar [B,n] and perm [B,n] (host; any may be NULL).
 * perm_by: 0 = permutation with the best mean SI-SDR, 1 = best mean SI-SIR (bss_eval's convention).
 * clamp_db <= 0: no clamping.  n <= 4. */
int dsn_si_bss_eval(dsn_ctx* ctx, const float* ref, const float* est, int B, int n, int L, int perm_by, float clamp_db,
                    float* si_sdr_out, float* si_sir_out, float* si_sar_out, int* perm_out, void* stream);

/* introspection for benchmarks / tests */
int dsn_enable_graphs(dsn_ctx* ctx, int enable);          /* hipGraph replay of sample/decode */
int64_t dsn_workspace_bytes(const dsn_ctx* ctx);
/* Per-launch HIP-event timing of the dominant (implicit-GEMM MFMA) kernel: between begin and
 * end every launch is bracketed by events on the launch stream (graphs are bypassed).
 * Returns summed kernel time, summed ALGORITHMIC flops (2*M*N*K of each contraction, the
 * 3x split-bf16 passes not counted) and the launch count. */
int dsn_profile_begin(dsn_ctx* ctx);
int dsn_profile_end(dsn_ctx* ctx, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches);
/* of the region closed by the last dsn_profile_end: the HBM-bound fused ResidualUnit launches alone (they are
 * also part of the totals above) -- summed kernel time, ALGORITHMIC bytes (operand planes in, fp32 residual in,
 * fp32 and planes out), launch count */
int dsn_profile_hbm(dsn_ctx* ctx, double* ms, double* bytes, int64_t* launches);
/* of the same region, per call site ("dit.qkv", "dit.ff_in", "dit.residual_norm", "vae.residual_unit_fused", ...):
 * summed event time, algorithmic flops and algorithmic HBM bytes (0 where not stated), launch count.  `names` is
 * max_rows x DSN_PROFILE_NAME_LEN chars.  Returns the number of rows available (may exceed max_rows). */
#define DSN_PROFILE_NAME_LEN 48
int dsn_profile_rows(dsn_ctx* ctx, int max_rows, char* names, double* ms, double* flops, double* bytes,
                     int64_t* launches);

/* ---- NOT PART OF THE ABI ------------------------------------------------------------------------------------
 * dsn_test_igemm, dsn_debug_read and dsn_bench_igemm below are hooks for this repository's own tests, repro scripts and
 * kernel sweeps.  They name internal workspace buffers and kernel variants, change without notice, and a binding of
 * the reference-facing interface must not use them. */
/* Test hook: run the implicit-GEMM kernel on caller-provided fp32 operands.
 * a [B][Lin][Cin] channels-last, w [N][taps*Cin]; out [B][rows_per_b][N] fp32 (no epilogue). */
int dsn_test_igemm(dsn_ctx* ctx, const float* a, const float* w, float* out, int B, int Lin, int Cin, int N,
                   int taps, int in_stride, int tap_dil, int in_pad, int rows_per_b, int panel_rows, int panel_bn,
                   void* stream);   /* panel_rows > 0: row-panel kernel with panel_bn (128|256) columns */

/* Development hook: copy `count` floats of a named workspace buffer to host memory. */
int dsn_debug_read(dsn_ctx* ctx, const char* name, float* host, int64_t count);
/* Development hook: average milliseconds of `iters` launches of the implicit-GEMM kernel on random
 * operands of the given contraction (variant 1 = register-staged core, 2 = glds-ring core). */
int dsn_bench_igemm(dsn_ctx* ctx, int B, int Lin, int Cin, int N, int taps, int tap_dil, int in_pad, int ksplit,
                    int variant, int iters, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* DITSEP_HIP_H */
