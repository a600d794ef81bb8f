// Shared device/host helpers for the gfx950 (MI355X) separation engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One 16-bit MFMA operand element, stored as raw bits: bf16 or IEEE fp16 depending on the
// engine's operand format (DSN_FMT_*).
typedef unsigned short op16_t;
typedef __attribute__((ext_vector_type(8))) unsigned short op16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short op16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DSN_WAVE 64

// ---- precision policy -----------------------------------------------------
// An activation / weight operand of an MFMA contraction is stored as P bf16
// "planes": plane 0 = bf16(v) (hi), plane 1 = bf16(v - hi) (lo).  P == 1 is the
// plain bf16 mode; P == 2 ("bf16x3") evaluates hi*hi + hi*lo + lo*hi on the
// bf16 matrix cores with fp32 accumulation: ~2^-16 relative operand error at
// 1/3 of the bf16 MFMA rate (gfx950 has no xf32, and f32-input MFMA runs at
// 1/16 of the bf16 rate).
// Operand format: bf16 (8-bit mantissa, fp32 range) or fp16 (11-bit mantissa: ~8x smaller
// rounding error than bf16 at the same MFMA rate; the reference trains under fp16 autocast).
// Kernels receive both packed in one int `pl`: plane count = pl & 3, fp16 flag = pl >> 4.
#define DSN_PL(planes, f16) ((planes) | ((f16) << 4))
#define PL_COUNT(pl) ((pl)&3)
#define PL_F16(pl) ((pl) >> 4)

// activations applied by producer epilogues when writing operand planes
enum { DSN_ACT_NONE = 0, DSN_ACT_ELU = 1, DSN_ACT_SNAKE = 2, DSN_ACT_SILU = 3 };
// transform of the fp32 output
enum { DSN_F32_NONE = 0, DSN_F32_TANH = 1 };

__device__ __forceinline__ float dsn_elu(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float dsn_silu(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float dsn_snake(float v, float alpha, float inv_beta) {
  float s = __sinf(v * alpha);
  return v + inv_beta * s * s;
}

// fp16 operands SATURATE at +-65504 (the largest finite fp16) instead of overflowing to inf: one out-of-range
// activation of a trained checkpoint (SwiGLU hidden units, Snake outputs) then costs its own clamping error, not a
// row of inf - inf = NaN downstream.  NaN stays NaN (a clamp that hid it would hide bugs).  bf16 has fp32's range.
#define DSN_F16_MAX 65504.f
__device__ __forceinline__ float dsn_sat16(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -DSN_F16_MAX, DSN_F16_MAX);
  return v != v ? v : c;
}
__device__ __forceinline__ op16_t to_op16(float v, int f16) {
  return f16 ? __builtin_bit_cast(unsigned short, (_Float16)dsn_sat16(v)) : __builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ float from_op16(op16_t u, int f16) {
  return f16 ? (float)__builtin_bit_cast(_Float16, u) : (float)__builtin_bit_cast(__bf16, u);
}
// hi = round(v), lo = round(v - hi) in the operand format
__device__ __forceinline__ void dsn_split(float v, op16_t& hi, op16_t& lo, int f16) {
  hi = to_op16(v, f16);
  lo = to_op16(v - from_op16(hi, f16), f16);
}

// ---- fp8 operands (DSN_PREC_FP8): OCP e4m3 data with one E8M0 scale byte per 32 consecutive K-elements of a row
// (the MX block format v_mfma_scale_f32_16x16x128_f8f6f4 consumes): value = fp8 * 2^(byte - 127).
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define DSN_FP8_MAX 448.f
// k with 2^k >= amax / 448 (so that amax * 2^-k fits e4m3), clamped to the normal-float exponent range
__device__ __forceinline__ int dsn_mx_exp(float amax) {
  const unsigned u = __builtin_bit_cast(unsigned, amax * (1.f / DSN_FP8_MAX));
  const int k = (int)((u >> 23) & 0xffu) - 127 + ((u & 0x7fffffu) ? 1 : 0);
  return min(max(k, -126), 126);
}
__device__ __forceinline__ float dsn_pow2(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }
// 4 floats -> 4 saturated e4m3 bytes (byte r = element r)
__device__ __forceinline__ unsigned dsn_fp8x4(const f32x4& v) {
  float c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float m = __builtin_amdgcn_fmed3f(v[r], -DSN_FP8_MAX, DSN_FP8_MAX);
    c[r] = v[r] != v[r] ? v[r] : m;
  }
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
  return (unsigned)w;
}
// one v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8): D[n][m] += sum_k w[n][k] a[m][k] 2^(sw-127) 2^(sa-127).
// Operand lane map (probed on gfx950, scripts/lab/mx_probe.hip): lane (r = lane & 15, q = lane >> 4) holds row r,
// bytes k = 16q .. 16q+15 in registers 0-3 and k = 64+16q .. 64+16q+15 in registers 4-7 -- the same two 16-byte
// chunks (q, 4 + q) of a 128-byte row that the two 16-bit k-steps of a 64-element tile read; its scale operand
// (byte 0) applies to the 32-element block q of row r.
__device__ __forceinline__ f32x4 mfma_mx8(const op16x8& w_lo, const op16x8& w_hi, const op16x8& a_lo, const op16x8& a_hi,
                                          const f32x4& c, int sw, int sa) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 wl = __builtin_bit_cast(i32x4, w_lo), wh = __builtin_bit_cast(i32x4, w_hi);
  const i32x4 al = __builtin_bit_cast(i32x4, a_lo), ah = __builtin_bit_cast(i32x4, a_hi);
  const i32x8 w = {wl[0], wl[1], wl[2], wl[3], wh[0], wh[1], wh[2], wh[3]};
  const i32x8 a = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0, 0, 0, sw, 0, sa);
}

// one v_mfma_f32_16x16x32 on raw 16-bit operand fragments: D[n][m] += sum_k w[n][k] a[m][k]
template <int F16>
__device__ __forceinline__ f32x4 mfma16(const op16x8& w, const op16x8& a, const f32x4& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0,
                                                   0);
}

// wave-wide reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time per-DEVICE setup from host launch code (function attributes, zero pages): several engine contexts on
// different GPUs may live in one process, so "done once per process" is not enough.
#include <atomic>
inline int dsn_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev & 63;
}
inline bool dsn_first_use_on_device(std::atomic<unsigned long long>& mask) {
  const unsigned long long bit = 1ull << dsn_current_device();
  if (mask.load(std::memory_order_relaxed) & bit) return false;
  mask.fetch_or(bit, std::memory_order_relaxed);
  return true;
}
