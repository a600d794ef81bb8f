// Implicit-GEMM MFMA kernel for gfx950 (see igemm.h for the contraction).
//
// Tile 128(m) x 128(n) x 32(k) per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators).  Operands are
// swapped on the matrix core: MFMA-A = weight rows (n), MFMA-B = activation
// rows (m), so each lane ends up holding FOUR CONSECUTIVE CHANNELS of one
// output row -> 16-byte fp32 / 8-byte bf16 channels-last stores and in-lane
// bias / activation / SwiGLU epilogues.
//
// K loop: taps outer, channel chunks inner; every (tap, chunk) stages a
// [128][32] activation tile whose rows are the tap-shifted input rows
// (zero-filled outside the sequence) and the matching weight tile.  Global ->
// register -> LDS staging, LDS double buffered, one barrier per k-tile: the
// loads of tile t+1 are issued before the MFMAs of tile t and written to the
// other buffer after them.
// LDS rows are 64 B; the 16-byte chunk index is XOR-swizzled with
// (-(row>>2))&3 so that every 16-lane ds_read_b128 group touches 16 distinct
// 16-byte slots of the 256-byte bank row.
#include "igemm.h"

#ifndef DSN_SETPRIO
#define DSN_SETPRIO 0
#endif
#ifndef DSN_SKINNY_U
#define DSN_SKINNY_U 4  // k-steps per load batch of the skinny kernel (development: 8 = the whole K share of a wave at K = 1024)
#endif
#ifndef DSN_DBG_MODE
#define DSN_DBG_MODE 0  // development ablation builds: 1 = no in-loop staging, 2 = no MFMAs, 3 = no epilogue (halo kernel)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_ELEMS = 128 * BK;  // one operand plane tile

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((-(row >> 2)) & 3); }

// Generic epilogue of one wave: NT column sub-tiles x MT row sub-tiles of 16x16 accumulators starting
// at (row mw0, column nw0); rows at or beyond m_end are not stored.
// ln_rows (LDS, folded-LayerNorm consumers): (mean, rstd) of panel row i at ln_rows[2 * i], i = m - ln_m0.
// EPI: compile-time feature bits, so that a kernel only carries (and allocates registers for) the epilogue code it can
// run: 1 = row statistics of the output (d.stat_out), 2 = folded-LayerNorm consumer (ln_rows), 4 = fp8 SwiGLU output.
enum { EPI_STATS = 1, EPI_LNFOLD = 2, EPI_FP8OUT = 4 };
// LEAN: feature groups compiled OUT of the generic epilogue (the launcher promises the descriptor does not use them):
// 1 = the DiT-only ones (RoPE, fused-QKV scaling, SwiGLU), 2 = the NCSN++-only ones (GroupNorm partials, per-item
// bias, tanh output).  The 16-wave 256 x 256 tile kernel (128-VGPR cap) spilled ~200 registers carrying all of them.
// 4 = the kernel seeded its accumulators with bias + per-item bias + residual before the k loop (seed_acc): the loads
// hide under the first k-tiles instead of forming a chain of dependent round trips in front of the stores.
enum { LEAN_NO_DIT = 1, LEAN_NO_NCSN = 2, LEAN_SEEDED = 4 };

// accumulator seed of one wave: bias[n] + bbias[b][n] + resid[m][n] for its NT x MT sub-tiles (rows >= m_end: zero)
template <int NT, int MT>
__device__ __forceinline__ void seed_acc(const GemmDesc& d, f32x4 (&acc)[NT][MT], int mw0, int m_end, int nw0, int lane) {
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
    const int m = mw0 + tm * 16 + (lane & 15);
    const bool mok = m < m_end;
    const int b = mok ? m / d.rows_per_b : 0;
    const int j = mok ? m - b * d.rows_per_b : 0;
    const long roff = (long)b * d.resid_bstride + (long)j * d.resid_row_elems + d.resid_off;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int n = nw0 + tn * 16 + nq;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (mok && n < d.N) {
        if (d.bias) v = *reinterpret_cast<const f32x4*>(d.bias + (n % d.bias_mod));
        if (d.sc_bias) v += *reinterpret_cast<const f32x4*>(d.sc_bias + n);
        if (d.bbias) v += *reinterpret_cast<const f32x4*>(d.bbias + (long)b * d.bbias_stride + n);
        if (d.resid) v += *reinterpret_cast<const f32x4*>(d.resid + roff + n);
      }
      acc[tn][tm] = v;
    }
  }
}
// "Touch" a loaded vector in uniform control flow: hipcc then places its s_waitcnt for the load HERE, once.  Left to
// the first use inside the per-row-sub-tile blocks (divergent: `if (m >= m_end) continue`), every block gets its own
// s_waitcnt vmcnt(0) -- the skipped path may still have the load in flight -- and since stores retire through the
// same counter, every block's stores wait for the previous block's to complete.
__device__ __forceinline__ void dsn_touch(const f32x4& v) { asm volatile("" ::"v"(v)); }

// Issue gap between the last MFMAs of a k loop and the epilogue's first VALU writes ("MFMA operand
// registers reused too early"): hipcc re-uses the A / B fragment registers of the final MFMAs for epilogue state within
// a few instructions of issuing them; experiment switch DSN_DRAIN_NOPS (0 = off).
#ifndef DSN_DRAIN_NOPS
#define DSN_DRAIN_NOPS 4
#endif
__device__ __forceinline__ void dsn_mfma_drain() {
#if DSN_DRAIN_NOPS > 0
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < DSN_DRAIN_NOPS; ++i) asm volatile("s_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// value of the lane ROT places to the right inside this lane's 16-lane row (wraps round the row)
template <int ROT>
__device__ __forceinline__ float dsn_row_ror(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x120 + ROT, 0xf, 0xf, false));
}

template <int P, int F16, int NT, int MT, int EPI = 0, int LEAN = 0>
__device__ __forceinline__ void epilogue_gen(const GemmDesc& d_arg, f32x4 (&acc)[NT][MT], int mw0, int m_end, int nw0,
                                             int lane, int z, const float* ln_rows = nullptr, int ln_m0 = 0) {
  dsn_mfma_drain();
  // A private copy (scalarised by the compiler): read through the kernel-argument reference, descriptor fields were
  // re-fetched after every output store -- the vector stores may alias anything as far as the compiler knows -- from a
  // scratch copy of the argument chunk, behind s_waitcnt vmcnt(0): every 16x16 tile's stores waited for the previous
  // tile's to complete.
  const GemmDesc d = d_arg;
  const bool f_rope = !(LEAN & LEAN_NO_DIT) && d.rope_cos != nullptr;
  const bool f_qkv = !(LEAN & LEAN_NO_DIT) && d.qkv_D > 0;
  const bool f_swiglu = !(LEAN & LEAN_NO_DIT) && d.swiglu;
#if DSN_DBG_MODE == 5
  const bool f_gn = false;
#else
  const bool f_gn = !(LEAN & LEAN_NO_NCSN) && d.gn_stats != nullptr;
#endif
  const bool f_bbias = !(LEAN & LEAN_NO_NCSN) && d.bbias != nullptr;
  const bool f_tanh = !(LEAN & LEAN_NO_NCSN) && d.f32_op == DSN_F32_TANH;
  const int nq = (lane >> 4) * 4;
  if (d.ksplit > 1) {  // raw partial sums to this slice's slab
    float* slab = d.out_f32 + (long)z * d.slab_stride;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const int m = mw0 + tm * 16 + (lane & 15);
      if (m >= m_end) continue;
      const int b = m / d.rows_per_b;
      const int j = m - b * d.rows_per_b;
      const long row_rel = (long)j * d.out_row_elems + d.out_off;
      const long row_abs = (long)b * d.out_bstride + row_rel;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n = nw0 + tn * 16 + nq;
        if (n >= d.N) continue;
        const long rel = row_rel + n;
        if (rel < 0 || rel >= d.out_limit) continue;
        *reinterpret_cast<f32x4*>(slab + row_abs + n) = acc[tn][tm];
      }
    }
    return;
  }
  if constexpr ((EPI & EPI_STATS) != 0) {
    // Residual-stream producer (to_out without split-K): x' = x + A W^T + bias, written in place as fp32, as the raw
    // operand plane, and summarised per row (mean, M2 of this wave's 64 columns).  Plain row-major [M][N] tensors.
    // The residual x is not read here: the kernel seeded the accumulators with it (C-in of the first MFMAs), so its
    // load latency hides under the first k-tiles and no load has to wait behind the in-place stores below.
    f32x4 bvec[NT];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int n = nw0 + tn * 16 + nq;
      bvec[tn] = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + (n % d.bias_mod)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) dsn_touch(bvec[tn]);
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const int m = mw0 + tm * 16 + (lane & 15);
      if (m >= m_end) continue;
      float s1 = 0.f;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n = nw0 + tn * 16 + nq;
        f32x4 v = acc[tn][tm] + bvec[tn];
        acc[tn][tm] = v;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        if (n < d.N) {
          *reinterpret_cast<f32x4*>(d.out_f32 + (long)m * d.N + n) = v;
          if constexpr ((EPI & EPI_FP8OUT) == 0) {
            op16x4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              op16_t h, l;
              dsn_split(v[r], h, l, F16);
              hi[r] = h;
              lo[r] = l;
            }
            *reinterpret_cast<op16x4*>(d.out_planes + (long)m * d.N + n) = hi;
            if (P == 2) *reinterpret_cast<op16x4*>(d.out_planes + d.out_ps + (long)m * d.N + n) = lo;
          }
        }
      }
      if constexpr ((EPI & EPI_FP8OUT) != 0) {
        // raw x' as the fp8 (MX) operand of the folded FF-in: a 32-column scale block = a pair of column sub-tiles of
        // this row across the 4 lane groups that share it (the partner lanes hold the same row: same branch)
#pragma unroll
        for (int tp = 0; tp < NT / 2; ++tp) {
          float amax = 0.f;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(acc[2 * tp + u][tm][r]));
          amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
          const int k = dsn_mx_exp(amax);
          const float inv = dsn_pow2(-k);
          const int nb = nw0 + tp * 32;
          if (nb < d.N) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
              *reinterpret_cast<unsigned*>(d.out_fp8 + (long)m * d.N + nb + u * 16 + nq) = dsn_fp8x4(acc[2 * tp + u][tm] * inv);
            if ((lane >> 4) == 0) d.out_fp8_scale[(long)m * (d.N >> 5) + (nb >> 5)] = (unsigned char)(k + 127);
          }
        }
      }
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = s1 * (1.f / (NT * 16));
      float m2 = 0.f;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dl = acc[tn][tm][r] - mean;
          m2 += dl * dl;
        }
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      if ((lane >> 4) == 0)
        *reinterpret_cast<float2*>(d.stat_out + ((long)m * d.stat_np + nw0 / (NT * 16)) * 2) = float2{mean, m2};
    }
    return;
  }
  if constexpr ((EPI & EPI_LNFOLD) != 0 && (EPI & EPI_FP8OUT) != 0) {
    // Folded-LayerNorm SwiGLU consumer with fp8 (MX) output: the wave's 64 packed columns are 32 output features = ONE
    // scale block of a row (see the unfolded fp8 SwiGLU branch below)
    static_assert(NT == 4, "fp8 SwiGLU epilogue expects 64 packed columns per wave");
    if (nw0 >= d.N) return;
    f32x4 cv[2], cg[2], bv[2], bg[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      const int np = nw0 + tp * 32;
      cv[tp] = *reinterpret_cast<const f32x4*>(d.ln_colsum + np + nq);
      cg[tp] = *reinterpret_cast<const f32x4*>(d.ln_colsum + np + 16 + nq);
      bv[tp] = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + np + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
      bg[tp] = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + np + 16 + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      dsn_touch(cv[tp]);
      dsn_touch(cg[tp]);
      dsn_touch(bv[tp]);
      dsn_touch(bg[tp]);
    }
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const int m = mw0 + tm * 16 + (lane & 15);
      if (m >= m_end) continue;  // (the lanes 16 / 32 / 48 away hold the same row: the shuffles below stay convergent)
      const float mu = ln_rows[2 * (m - ln_m0)], rs = ln_rows[2 * (m - ln_m0) + 1];
      f32x4 h[2];
      float amax = 0.f;
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const f32x4 val = (acc[2 * tp][tm] - cv[tp] * mu) * rs + bv[tp];
        const f32x4 gate = (acc[2 * tp + 1][tm] - cg[tp] * mu) * rs + bg[tp];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[tp][r] = val[r] * dsn_silu(gate[r]);
          amax = fmaxf(amax, fabsf(h[tp][r]));
        }
      }
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const int k = dsn_mx_exp(amax);
      const float inv = dsn_pow2(-k);
      const long orow = (long)m * (d.N >> 1);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
        *reinterpret_cast<unsigned*>(d.out_fp8 + orow + (nw0 >> 1) + tp * 16 + nq) = dsn_fp8x4(h[tp] * inv);
      if ((lane >> 4) == 0) d.out_fp8_scale[(long)m * (d.N >> 6) + (nw0 >> 6)] = (unsigned char)(k + 127);
    }
    return;
  }
  if constexpr ((EPI & EPI_LNFOLD) != 0 && (EPI & EPI_FP8OUT) == 0) {
    // Folded-LayerNorm SwiGLU consumer: LN(x') W^T = rstd (x' (W diag gamma)^T - mean colsum) + bias', then
    // value * silu(gate) -> operand planes [M][N/2].  Column vectors are loaded once, rows then stream out.
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      const int np = nw0 + tp * 32;
      if (np >= d.N) continue;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(d.ln_colsum + np + nq);
      const f32x4 cg = *reinterpret_cast<const f32x4*>(d.ln_colsum + np + 16 + nq);
      const f32x4 bv = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + np + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 bg = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + np + 16 + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
      dsn_touch(cv);
      dsn_touch(cg);
      dsn_touch(bv);
      dsn_touch(bg);
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        const int m = mw0 + tm * 16 + (lane & 15);
        if (m >= m_end) continue;
        const float mu = ln_rows[2 * (m - ln_m0)], rs = ln_rows[2 * (m - ln_m0) + 1];
        const f32x4 val = (acc[2 * tp][tm] - cv * mu) * rs + bv;
        const f32x4 gate = (acc[2 * tp + 1][tm] - cg * mu) * rs + bg;
        op16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          op16_t h, l;
          dsn_split(val[r] * dsn_silu(gate[r]), h, l, F16);
          hi[r] = h;
          lo[r] = l;
        }
        const long off = (long)m * (d.N >> 1) + (np >> 1) + nq;
        *reinterpret_cast<op16x4*>(d.out_planes + off) = hi;
        if (P == 2) *reinterpret_cast<op16x4*>(d.out_planes + d.out_ps + off) = lo;
      }
    }
    return;
  }
  // Everything the store loop would LOAD is fetched here, in uniform control flow (see dsn_touch): the rotary tables
  // (applied right away, to every row sub-tile) and the bias vectors of the wave's column sub-tiles.
  constexpr bool ROPE_FIRST = MT <= 2;  // taller wave tiles have no registers for it (they spill): rotary per sub-tile
  if (ROPE_FIRST && !f_swiglu && f_rope) {
    // fused QKV epilogue: this wave's 64 columns are one 64-wide head of the q, k or v section; rotary embedding on the
    // first 32 features (pairs (c, c+16) = accumulator tiles 0 and 1 of the same lane), q pre-scaled by 1/sqrt(dh)
    const int section = nw0 / d.qkv_D;
    if (section < 2 && (nw0 & 63) == 0) {  // rotary features live in the first two 16-tiles of a head
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        const int pos = (mw0 + tm * 16 + (lane & 15)) % d.rope_S;  // rows past m_end: any table row, never stored
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + nq);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + nq);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + 16 + nq);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + 16 + nq);
        const f32x4 x0 = acc[0][tm], x1 = acc[1][tm];
        acc[0][tm] = x0 * c0 - x1 * s0;
        acc[1][tm] = x1 * c1 + x0 * s1;
      }
    }
  }
  constexpr bool BIAS_FIRST = MT <= 4;  // (taller wave tiles -- the 13- and 17-sub-tile panels -- would spill)
  f32x4 gbias[NT];
  if constexpr ((LEAN & LEAN_SEEDED) == 0 && BIAS_FIRST) {
    if (!f_swiglu) {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n = nw0 + tn * 16 + nq;
        gbias[tn] = (d.bias && n < d.N) ? *reinterpret_cast<const f32x4*>(d.bias + (n % d.bias_mod)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) dsn_touch(gbias[tn]);
    }
  }
  // GroupNorm slice partials in ONE pass over the accumulators, so that a sub-tile's registers are free once it is stored
  // (a second, deviation pass kept all 64 alive across the store loop: 60 spilled registers in the 128-VGPR halo kernel).
  // Per lane and column sub-tile a running (mean, M2) pair: every row sub-tile contributes the exact two-pass (mean, M2)
  // of its 4 values through Chan's update -- no E[x^2] - E[x]^2, no shift value to carry; the 16 row lanes are merged
  // with the same formula over a fixed rotation tree below.
  float gmean[NT], gm2[NT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) gmean[tn] = gm2[tn] = 0.f;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
    const int m = mw0 + tm * 16 + (lane & 15);
    if (m >= m_end) continue;
    const int b = m / d.rows_per_b;
    const int j = m - b * d.rows_per_b;
    const long row_rel = (long)j * d.out_row_elems + d.out_off;
    const long row_abs = (long)b * d.out_bstride + row_rel;
    if (!f_swiglu) {
      if (!ROPE_FIRST && f_rope) {
        const int section = nw0 / d.qkv_D;
        if (section < 2 && (nw0 & 63) == 0) {
          const int pos = m % d.rope_S;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + nq);
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + nq);
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + 16 + nq);
          const f32x4 s1 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + 16 + nq);
          const f32x4 x0 = acc[0][tm], x1 = acc[1][tm];
          acc[0][tm] = x0 * c0 - x1 * s0;
          acc[1][tm] = x1 * c1 + x0 * s1;
        }
      }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n = nw0 + tn * 16 + nq;
        if (n >= d.N) continue;
        const long rel = row_rel + n;
        if (rel < 0 || rel >= d.out_limit) continue;
        const long off = row_abs + n;
        f32x4 v = acc[tn][tm];
        if constexpr ((LEAN & LEAN_SEEDED) == 0) {
          if constexpr (BIAS_FIRST) v += gbias[tn];
          else if (d.bias) v += *reinterpret_cast<const f32x4*>(d.bias + (n % d.bias_mod));
          if (f_qkv && n < d.qkv_D) v *= d.q_scale;  // fused q|k|v projection: q (bias included) pre-scaled
          if (f_bbias) v += *reinterpret_cast<const f32x4*>(d.bbias + (long)b * d.bbias_stride + n);
          if (d.resid)
            v += *reinterpret_cast<const f32x4*>(d.resid + (long)b * d.resid_bstride + (long)j * d.resid_row_elems +
                                                 d.resid_off + n);
        }
        v *= d.out_scale;
        if (f_gn) {
          // exact (mean, M2) of the lane's 4 values, merged into the running pair of this column sub-tile with Chan's
          // update (counts are compile-time constants: 4 tm values so far, 4 new)
          const float mb = 0.25f * ((v[0] + v[1]) + (v[2] + v[3]));
          float m2b = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dl = v[r] - mb;
            m2b += dl * dl;
          }
          if (tm == 0) {
            gmean[tn] = mb;
            gm2[tn] = m2b;
          } else {
            const float dm = mb - gmean[tn];
            gmean[tn] += dm * (1.f / (float)(tm + 1));
            gm2[tn] += m2b + dm * dm * (4.f * (float)tm / (float)(tm + 1));
          }
        }
        if (d.out_f32) {
          f32x4 o = v;
          if (f_tanh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = tanhf(v[r]);
          }
#if DSN_DBG_MODE == 4
          asm volatile("" ::"v"(o));
#else
          *reinterpret_cast<f32x4*>(d.out_f32 + off) = o;
#endif
        }
        if (d.out_planes) {
          f32x4 a = v;
          if (d.act == DSN_ACT_ELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = dsn_elu(v[r]);
          } else if (d.act == DSN_ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = dsn_silu(v[r]);
          } else if (d.act == DSN_ACT_SNAKE) {
            const int ch = n % d.act_mod;
            const f32x4 al = *reinterpret_cast<const f32x4*>(d.act_a + ch);
            const f32x4 ib = *reinterpret_cast<const f32x4*>(d.act_b + ch);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = dsn_snake(v[r], al[r], ib[r]);
          }
          op16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            op16_t h, l;
            dsn_split(a[r], h, l, F16);
            hi[r] = h;
            lo[r] = l;
          }
          *reinterpret_cast<op16x4*>(d.out_planes + off) = hi;
          if (P == 2) *reinterpret_cast<op16x4*>(d.out_planes + d.out_ps + off) = lo;
        }
      }
    } else if ((EPI & EPI_FP8OUT) && d.out_fp8) {
      // SwiGLU with fp8 (MX) output: the wave's 64 packed columns are 32 output features = ONE scale block of this
      // row: amax over the lane's 8 values and the 4 lane groups that share the row, E8M0 scale, saturated e4m3
      static_assert(!(EPI & EPI_FP8OUT) || NT == 4, "fp8 SwiGLU epilogue expects 64 packed columns per wave");
      f32x4 h[2];
      float amax = 0.f;
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int np = nw0 + tp * 32;
        f32x4 val = acc[2 * tp][tm], gate = acc[2 * tp + 1][tm];
        if (d.bias && np < d.N) {
          val += *reinterpret_cast<const f32x4*>(d.bias + np + nq);
          gate += *reinterpret_cast<const f32x4*>(d.bias + np + 16 + nq);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[tp][r] = val[r] * dsn_silu(gate[r]);
          amax = fmaxf(amax, fabsf(h[tp][r]));
        }
      }
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const int k = dsn_mx_exp(amax);
      const float inv = dsn_pow2(-k);
      if (nw0 < d.N) {
        const long orow = (long)m * (d.N >> 1);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
          *reinterpret_cast<unsigned*>(d.out_fp8 + orow + (nw0 >> 1) + tp * 16 + nq) = dsn_fp8x4(h[tp] * inv);
        if ((lane >> 4) == 0) d.out_fp8_scale[(long)m * (d.N >> 6) + (nw0 >> 6)] = (unsigned char)(k + 127);
      }
    } else {
      // SwiGLU: packed rows [32g, 32g+16) = value features 16g.., [32g+16, 32g+32) = their gates
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        const int np = nw0 + tp * 32;
        if (np >= d.N) continue;
        const int feat = (np >> 1) + nq;
        const long off = row_abs + feat;
        f32x4 val = acc[2 * tp][tm], gate = acc[2 * tp + 1][tm];
        if (d.bias) {
          val += *reinterpret_cast<const f32x4*>(d.bias + np + nq);
          gate += *reinterpret_cast<const f32x4*>(d.bias + np + 16 + nq);
        }
        op16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          op16_t h, l;
          dsn_split(val[r] * dsn_silu(gate[r]), h, l, F16);
          hi[r] = h;
          lo[r] = l;
        }
        *reinterpret_cast<op16x4*>(d.out_planes + off) = hi;
        if (P == 2) *reinterpret_cast<op16x4*>(d.out_planes + d.out_ps + off) = lo;
      }
    }
  }
  if (f_gn && mw0 < m_end) {
    // the wave's rows are one whole 64-row slice of one item (rows_per_b % 64 == 0, so no row of it is masked): one
    // plain store per (slice, quad) -- no atomics, every launch writes the same bits
    const int b = mw0 / d.rows_per_b;
    const int slice = (mw0 - b * d.rows_per_b) >> 6;
    const int S = d.rows_per_b >> 6;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      float cnt = (float)(MT * 4);
      float mean = gmean[tn], m2 = gm2[tn];
      // equal-count merge over the 16 row lanes: mean' = (mA + mB)/2, M2' = M2A + M2B + (mB - mA)^2 n/2, partner =
      // the lane 8, 4, 2, 1 places round the 16-lane row (DPP row rotate: registers only, no LDS-unit permute)
#define DSN_GN_MERGE(ROT)                                        \
      {                                                          \
        const float om = dsn_row_ror<ROT>(mean), o2 = dsn_row_ror<ROT>(m2); \
        const float dm = om - mean;                              \
        m2 = m2 + o2 + dm * dm * (0.5f * cnt);                   \
        mean = 0.5f * (mean + om);                               \
        cnt *= 2.f;                                              \
      }
      DSN_GN_MERGE(8) DSN_GN_MERGE(4) DSN_GN_MERGE(2) DSN_GN_MERGE(1)
#undef DSN_GN_MERGE
      const int n = nw0 + tn * 16 + nq;
      if ((lane & 15) == 0 && n < d.N) {
        if (!(LEAN & LEAN_NO_NCSN) && d.gnf_out) {
          // read again in THIS kernel by the other row tiles of the image (igemm_halo3x3_kernel): write-through stores
          float* sp = d.gn_stats + ((((long)b * S + slice) * (d.N >> 2)) + (n >> 2)) * 2;
          __hip_atomic_store(sp, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp + 1, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
        *reinterpret_cast<float2*>(d.gn_stats + ((((long)b * S + slice) * (d.N >> 2)) + (n >> 2)) * 2) = float2{mean, m2};
        if (d.gn_stats2)
          *reinterpret_cast<float2*>(d.gn_stats2 + ((((long)b * S + slice) * d.gn_nq2) + d.gn_qoff2 + (n >> 2)) * 2) =
              float2{mean, m2};
      }
    }
  }
}

template <int P, int F16, int LEAN = 0>
__device__ __forceinline__ void epilogue_tile(const GemmDesc& d, f32x4 (&acc)[4][4], int mw0, int nw0, int lane,
                                              int z) {
  epilogue_gen<P, F16, 4, 4, 0, LEAN>(d, acc, mw0, d.M, nw0, lane, z);
}

template <int P, int F16>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const GemmDesc d) {
  // [buf][operand A=0/W=1][plane][row*32 + chunk*8]
  __shared__ __attribute__((aligned(16))) op16_t lds[2 * 2 * P * TILE_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware, bijective tile remap: blocks b and b+8 share an XCD (L2); give
  // each XCD a contiguous run of tiles with tile_n fastest so the blocks that
  // re-read one activation row panel sit behind the same L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tile_m = t / d.tiles_n, tile_n = t - tile_m * d.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int Ktot = d.taps * d.Cin;
  const int kc_per_tap = d.Cin / BK;
  const int nkt = d.taps * kc_per_tap;

  // ---- staging assignment: 2 chunks of A and 2 of W per plane per thread ----
  int s_row[2], s_kch[2], s_lds[2];
  long a_base[2];
  int a_js[2];
  bool a_ok[2], w_ok[2];
  long w_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
    const int row = c >> 2, kch = c & 3;
    s_row[i] = row;
    s_kch[i] = kch;
    s_lds[i] = row * BK + swz(row, kch) * 8;
    const int m = m0 + row;
    const int b = m / d.rows_per_b;
    const int j = m - b * d.rows_per_b;
    a_ok[i] = m < d.M;
    a_base[i] = (long)b * d.in_bstride + kch * 8;
    a_js[i] = j * d.in_stride - d.in_pad;
    const int n = n0 + row;
    w_ok[i] = n < d.N;
    w_base[i] = (long)n * Ktot + kch * 8;
  }

  op16x8 ra[P][2], rw[P][2];
  const op16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto stage_load = [&](int kt) {
    const int tap = kt / kc_per_tap;
    const int kc = kt - tap * kc_per_tap;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = a_js[i] + tap * d.tap_dil;
      bool ok = a_ok[i] && r >= 0 && r < d.Lin;
      if (d.img_w > 0) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int y = a_js[i] / d.img_w, x = a_js[i] - y * d.img_w;
        ok = a_ok[i] && (unsigned)(y + dy) < (unsigned)d.img_h && (unsigned)(x + dx) < (unsigned)d.img_w;
        r = a_js[i] + dy * d.img_w + dx;
      }
      const long aoff = a_base[i] + (long)r * d.in_row_elems + kc * BK;
      const long woff = w_base[i] + (long)tap * d.Cin + kc * BK;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        ra[p][i] = ok ? *reinterpret_cast<const op16x8*>(d.A + p * d.a_ps + aoff) : zero8;
        rw[p][i] = w_ok[i] ? *reinterpret_cast<const op16x8*>(d.W + p * d.w_ps + woff) : zero8;
      }
    }
  };
  auto stage_store = [&](int buf) {
    op16_t* base = lds + buf * (2 * P * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        *reinterpret_cast<op16x8*>(base + (0 * P + p) * TILE_ELEMS + s_lds[i]) = ra[p][i];
        *reinterpret_cast<op16x8*>(base + (1 * P + p) * TILE_ELEMS + s_lds[i]) = rw[p][i];
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (elements) inside a plane tile
  const int frow = lane & 15, fchunk = lane >> 4;
  const int fsw = swz(frow, fchunk) * 8;  // (tile*16 + wave offset) is a multiple of 16 -> swizzle depends on frow only
  const int a_frag_off = (wm * 64 + frow) * BK + fsw;
  const int w_frag_off = (wn * 64 + frow) * BK + fsw;

  stage_load(0);
  stage_store(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage_load(kt + 1);

    const op16_t* base = lds + buf * (2 * P * TILE_ELEMS);
    op16x8 fa[P][4], fw[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[p][i] = *reinterpret_cast<const op16x8*>(base + (0 * P + p) * TILE_ELEMS + a_frag_off + i * 16 * BK);
        fw[p][i] = *reinterpret_cast<const op16x8*>(base + (1 * P + p) * TILE_ELEMS + w_frag_off + i * 16 * BK);
      }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        if (P == 2) {
          acc[tn][tm] = mfma16<F16>(fw[P - 1][tn], fa[0][tm], acc[tn][tm]);
          acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[P - 1][tm], acc[tn][tm]);
        }
        acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[0][tm], acc[tn][tm]);
      }
    }
    if (kt + 1 < nkt) stage_store(buf ^ 1);
    __syncthreads();
  }

  epilogue_tile<P, F16>(d, acc, m0 + wm * 64, n0 + wn * 64, lane, 0);
}

}  // namespace

hipError_t igemm_launch(const GemmDesc& din, int pl, hipStream_t stream) {
  const int planes = PL_COUNT(pl), f16 = PL_F16(pl);
  GemmDesc d = din;
  d.tiles_m = cdiv(d.M, BM);
  d.tiles_n = cdiv(d.N, BN);
  if (d.Cin % BK != 0 || d.M <= 0 || d.N <= 0) return hipErrorInvalidValue;
  if (d.swiglu && (d.N % 32 != 0)) return hipErrorInvalidValue;
  const int grid = d.tiles_m * d.tiles_n;
  if (planes == 1 && !f16)
    hipLaunchKernelGGL((igemm_kernel<1, 0>), dim3(grid), dim3(256), 0, stream, d);
  else if (planes == 2 && !f16)
    hipLaunchKernelGGL((igemm_kernel<2, 0>), dim3(grid), dim3(256), 0, stream, d);
  else if (planes == 1)
    hipLaunchKernelGGL((igemm_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, d);
  else
    hipLaunchKernelGGL((igemm_kernel<2, 1>), dim3(grid), dim3(256), 0, stream, d);
  return hipGetLastError();
}

// ============================================================================
// v2 core: direct-to-LDS staging (global_load_lds_dwordx4) into an NSTAGE ring,
// NSTAGE-1 k-tiles in flight behind a counted s_waitcnt vmcnt(N) and ONE raw
// s_barrier per k-tile; workgroup tile BM x BN in {128,256}^2 built from 64x64
// wave tiles (4 / 8 / 16 waves); optional split-K.
//
// glds writes LDS lane-linearly (wave base + lane*16 B): one wave-instruction
// fills a 16-row x 64-B group of a plane tile, lane l -> row l>>2, chunk slot
// l&3.  The bank swizzle therefore goes on the SOURCE address (slot c holds global
// chunk c ^ f(row)) and on the fragment read, never on the destination.  Rows that
// fall outside the sequence (conv zero padding, M/N tails) read a zero page.
// ============================================================================
namespace {

// Loader state of one staged row (one 16-byte chunk slot of it) for the glds kernels: the row's base
// pointer per plane (tap 0, channel chunk 0) and a bitmask of the taps for which the row exists
// (conv zero padding / M,N tails -> zero page).  Everything that varies per k-tile is wave-uniform:
// a scalar element offset added to the base.
struct RowLoad {
  const op16_t* ptr;  // plane 0; plane p adds p * ps
  long ps;
  unsigned mask;
  bool is_a;
};

__device__ __forceinline__ RowLoad make_row(const GemmDesc& d, bool is_a, int m_or_n, bool in_range, int gchunk, int Ktot) {
  RowLoad r;
  r.is_a = is_a;
  if (is_a) {
    const int b = m_or_n / d.rows_per_b;
    const int j = m_or_n - b * d.rows_per_b;
    const int js = j * d.in_stride - d.in_pad;
    r.ptr = d.A + (long)b * d.in_bstride + (long)js * d.in_row_elems + gchunk * 8;
    r.ps = d.a_ps;
    unsigned mk = 0;
    if (in_range) {
      if (d.img_w > 0) {
        const int y = j / d.img_w, x = j - y * d.img_w;
        for (int t = 0; t < d.taps; ++t) {
          const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
          if ((unsigned)(y + dy) < (unsigned)d.img_h && (unsigned)(x + dx) < (unsigned)d.img_w) mk |= 1u << t;
        }
      } else {
        for (int t = 0; t < d.taps; ++t)
          if ((unsigned)(js + t * d.tap_dil) < (unsigned)d.Lin) mk |= 1u << t;
      }
    }
    r.mask = mk;
  } else {
    r.ptr = d.W + (long)m_or_n * Ktot + gchunk * 8;
    r.ps = d.w_ps;
    r.mask = in_range ? 0xffffffffu : 0u;
  }
  return r;
}

// wave-uniform element offsets of k-tile (tap, kc): activation rows / weight rows
__device__ __forceinline__ long a_tile_off(const GemmDesc& d, int tap, int kc, int bk) {
  const int roff = d.img_w > 0 ? (tap / 3 - 1) * d.img_w + (tap - (tap / 3) * 3 - 1) : tap * d.tap_dil;
  return (long)roff * d.in_row_elems + kc * bk;
}
__device__ __forceinline__ long w_tile_off(const GemmDesc& d, int tap, int kc, int bk) {
  return (long)tap * d.Cin + kc * bk;
}

// swizzle of the 16-byte chunk index inside a TBK-wide LDS row (conflict-free ds_read_b128 of
// MFMA fragments: rows 64 B -> 4 rows per 256-B bank window, rows 128 B -> 2 rows per window)
template <int TBK>
__device__ __forceinline__ int swzk(int row) {
  return TBK == 32 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7);
}

// WTN: columns of a wave tile (64, or 32 for the 128 x 64 workgroup tile that gives a GEMM of few rows twice the workgroups)
template <int P, int F16, int TBM, int TBN, int NST, int TBK, int LEAN = 0, int WTN = 64>
__global__ __launch_bounds__((TBM / 64) * (TBN / WTN) * 64, 1) void igemm2_kernel(const GemmDesc d,
                                                                                  const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];  // [NST][plane][A rows | W rows][TBK]
  constexpr int WN_ = TBN / WTN;
  constexpr int NT = WTN / 16;
  constexpr int NWAVES = (TBM / 64) * WN_;
  constexpr int ROWS = TBM + TBN;             // staged rows per plane per k-tile (A rows then W rows)
  constexpr int PLANE_ELEMS = ROWS * TBK;
  constexpr int STAGE_ELEMS = P * PLANE_ELEMS;
  constexpr int CPR = TBK / 8;                // 16-byte chunks per row
  constexpr int RPG = 64 / CPR;               // rows filled by one glds wave-instruction (1 KiB)
  constexpr int GROUPS = ROWS / RPG;
  constexpr int GPW = GROUPS / NWAVES;        // groups per wave
  static_assert(GROUPS % NWAVES == 0, "row groups must divide over the waves");
  constexpr int G = GPW * P;                  // glds wave-instructions per wave per k-tile
  constexpr int KS = TBK / 32;                // MFMA k-steps per k-tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave - wm * WN_;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int ntiles = d.tiles_m * d.tiles_n;
  const int z = t / ntiles;
  const int tile = t - z * ntiles;
  // m_fast: consecutive tiles (one XCD's share) walk DOWN the rows of a few column tiles, so the XCD's
  // L2 keeps its weight columns and streams the (small) activation panels -- for GEMMs with few rows and
  // many columns (DiT); default walks ACROSS the columns of a few row panels (conv stacks: huge M, small N).
  const int tile_m = d.m_fast ? tile % d.tiles_m : tile / d.tiles_n;
  const int tile_n = d.m_fast ? tile / d.tiles_m : tile - tile_m * d.tiles_n;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  const int Ktot = d.taps * d.Cin;
  const int kc_per_tap = d.Cin / TBK;
  const int nkt_all = d.taps * kc_per_tap;
  const int kt_begin = (int)((long)nkt_all * z / d.ksplit);
  const int kt_end = (int)((long)nkt_all * (z + 1) / d.ksplit);
  const int nkt = kt_end - kt_begin;

  // ---- loader role: this wave stages row groups [wave*GPW, wave*GPW + GPW) ----
  const int rsub = lane / CPR, cpos = lane % CPR;
  RowLoad rl[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave * GPW + gi;
    const bool is_a = g < TBM / RPG;
    const int row = (is_a ? g : g - TBM / RPG) * RPG + rsub;
    const int gchunk = cpos ^ swzk<TBK>(row);  // source chunk held by this lane's LDS slot
    const int idx = (is_a ? m0 : n0) + row;
    rl[gi] = make_row(d, is_a, idx, is_a ? idx < d.M : idx < d.N, gchunk, Ktot);
  }
  const op16_t* zsrc = zero_page + cpos * 8;
  int itap = kt_begin / kc_per_tap, ikc = kt_begin - itap * kc_per_tap;  // (tap, chunk) of the next tile to issue

  auto issue = [&](int stage) {
    op16_t* sbase = lds + stage * STAGE_ELEMS + wave * GPW * RPG * TBK;
    const long offa = a_tile_off(d, itap, ikc, TBK), offw = w_tile_off(d, itap, ikc, TBK);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const bool ok = (rl[gi].mask >> itap) & 1u;
      const op16_t* g0 = rl[gi].ptr + (rl[gi].is_a ? offa : offw);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const op16_t* g = ok ? g0 + p * rl[gi].ps : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(sbase + p * PLANE_ELEMS + gi * RPG * TBK),
                                         16, 0, 0);
      }
    }
    if (++ikc == kc_per_tap) {
      ikc = 0;
      ++itap;
    }
  };

  f32x4 acc[NT][4];
  if constexpr ((LEAN & LEAN_SEEDED) != 0) {
    seed_acc<NT, 4>(d, acc, m0 + wm * 64, d.M, n0 + wn * WTN, lane);  // issued before any glds: lands first (in order)
  } else {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // fragment rows are (tile*16 + frow) with tile offsets multiples of 16: the swizzle only sees frow
  const int frow = lane & 15, fchunk = lane >> 4;
  const int fsw = swzk<TBK>(frow);
  const int a_row_off = (wm * 64 + frow) * TBK;
  const int w_row_off = (TBM + wn * WTN + frow) * TBK;

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) issue(s);

  for (int i = 0; i < nkt; ++i) {
    // tile i has landed once at most the loads of the NST-2 younger tiles are outstanding; lgkmcnt(0): this wave's
    // fragment reads of tile i-1 have completed (not merely been issued) before the barrier lets another wave's
    // glds overwrite that stage -- hipcc will otherwise leave reads in flight across the barrier
    const int younger = min(NST - 2, nkt - 1 - i);
    if (NST >= 4 && younger >= 2)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * G) : "memory");
    else if (NST >= 3 && younger >= 1)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's part of tile i landed; everyone finished tile i-1
#if DSN_DBG_MODE != 1
    if (i + NST - 1 < nkt) issue((i + NST - 1) % NST);
#endif

    const op16_t* base = lds + (i % NST) * STAGE_ELEMS;
#if DSN_DBG_MODE == 2
    {
      op16x8 v = *reinterpret_cast<const op16x8*>(base + a_row_off);
      asm volatile("" ::"v"(v));
      continue;
    }
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = ((ks * 4 + fchunk) ^ fsw) * 8;
      op16x8 fa[P][4], fw[P][NT];
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          fa[p][k] = *reinterpret_cast<const op16x8*>(base + p * PLANE_ELEMS + a_row_off + k * 16 * TBK + coff);
#pragma unroll
        for (int k = 0; k < NT; ++k)
          fw[p][k] = *reinterpret_cast<const op16x8*>(base + p * PLANE_ELEMS + w_row_off + k * 16 * TBK + coff);
      }
#if DSN_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          if (P == 2) {
            acc[tn][tm] = mfma16<F16>(fw[P - 1][tn], fa[0][tm], acc[tn][tm]);
            acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[P - 1][tm], acc[tn][tm]);
          }
          acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[0][tm], acc[tn][tm]);
        }
      }
#if DSN_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  }
  epilogue_gen<P, F16, NT, 4, 0, LEAN>(d, acc, m0 + wm * 64, d.M, n0 + wn * WTN, lane, z);
  if constexpr (P == 1 && TBM == 128 && TBN == 64 && WTN == 32 && LEAN == 0) {
    if (d.gnf_out) {
      // ---- GroupNorm finished by this workgroup alone (GemmDesc::gnf_out with 128-pixel images: the tile IS the
      // image, its 64 columns hold whole groups): the epilogue above stored nothing but the two slices' partials; they are
      // combined exactly as gn_apply_kernel does and silu(GroupNorm(out)) goes to the next conv's operand plane from the
      // accumulators.  No other workgroup is involved (igemm_halo3x3_kernel has the cross-workgroup form).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float* const gm = reinterpret_cast<float*>(lds);
      float* const gr = gm + 64;
      const int HW = d.rows_per_b, b = m0 / HW;
      const int C = d.N, nq_all = C >> 2;
      const int G = min(C >> 2, 32), cpg = C / G, qpg = cpg >> 2;
      const int S = HW >> 6;
      int tpg = 1;
      while (tpg < 64 && G * tpg * 2 <= 256) tpg *= 2;
      const int ngt = min(TBN, C - n0) / cpg;
      {
        const int gl = tid / tpg, sub = tid - gl * tpg;
        const bool live = gl < ngt;
        const float* sp = d.gn_stats + ((long)b * S * nq_all + (long)(n0 / cpg + (live ? gl : 0)) * qpg) * 2;
        const int items = S * qpg;
        const float cnt = 256.f;
        float wsum = 0.f;
        if (live)
          for (int it = sub; it < items; it += tpg) {
            const int sl = it / qpg, q2 = it - sl * qpg;
            wsum += cnt * __hip_atomic_load(sp + ((long)sl * nq_all + q2) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        for (int o = tpg >> 1; o >= 1; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
        const float ntot = (float)HW * (float)cpg;
        const float mean = wsum / ntot;
        float m2 = 0.f;
        if (live)
          for (int it = sub; it < items; it += tpg) {
            const int sl = it / qpg, q2 = it - sl * qpg;
            const float* pp = sp + ((long)sl * nq_all + q2) * 2;
            const float px = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float py = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float dm = px - mean;
            m2 += py + cnt * dm * dm;
          }
        for (int o = tpg >> 1; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        if (live && sub == 0) {
          gm[gl] = mean;
          gr[gl] = rsqrtf(m2 / ntot + d.gnf_eps);
        }
      }
      __syncthreads();
      const int nqc = (lane >> 4) * 4;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n = n0 + wn * WTN + tn * 16 + nqc;
        if (n >= C) continue;
        const int gl = (n - n0) / cpg;
        const float mean = gm[gl], rstd = gr[gl];
        const f32x4 ga = *reinterpret_cast<const f32x4*>(d.gnf_gamma + n);
        const f32x4 be = *reinterpret_cast<const f32x4*>(d.gnf_beta + n);
        const f32x4 bias4 = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + (n % d.bias_mod)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          const int m = m0 + wm * 64 + tm * 16 + (lane & 15);
          if (m >= d.M) continue;
          // the value the epilogue would have stored: (acc + bias) + per-item bias, scaled (same order of operations)
          f32x4 v = acc[tn][tm] + bias4;
          if (d.bbias) v += *reinterpret_cast<const f32x4*>(d.bbias + (long)b * d.bbias_stride + n);
          op16x4 h;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float vk = v[k] * d.out_scale;
            asm volatile("" : "+v"(vk));  // (rounded on its own, as the stored fp32 tensor would hold it)
            float tt = (vk - mean) * rstd * ga[k] + be[k];
            h[k] = to_op16(d.gnf_silu ? dsn_silu(tt) : tt, F16);
          }
          *reinterpret_cast<op16x4*>(d.gnf_out + (long)m * C + n) = h;
        }
      }
    }
  }
}

// ============================================================================
// Halo-resident 3x3 convolution (NCSN++ `ddpm_conv3x3` on channels-last images [B][H*W][C], single-plane modes).
// igemm2_kernel stages the activation tile once per tap (9 x per channel chunk).  Here a workgroup owns TBM
// consecutive rows of ONE image (TBM | H*W) x 128 output channels and, per 32-channel chunk, stages the rows
// [first - (W+1), last + (W+1)] ONCE (double buffered) -- every tap (dy, dx) is the same LDS chunk read at a row shift
// of dy*W + dx -- while the weights stream through their own 4-slot ring, one (tap, chunk) tile per k-step.
// Staged bytes per workgroup at 256 x 128: 0.38 MB instead of 0.89 MB (Cin = 128).
//   * rows beyond the image (y = -1, y = H) are zero-page rows of the halo;
//   * the x borders are the only taps that read a VALID neighbour which must count as zero (x = 0 with dx = -1
//     reads the previous image row's last pixel): those A fragments are zeroed in registers, per lane;
//   * the 64-byte-row swizzle of igemm2 (chunk ^ (-(row >> 2) & 3)) is conflict-free for any row shift: 16 consecutive
//     rows never hold two rows 16 apart, and rows of the partial first / last 4-row group differ in row % 4.
// Ring protocol as everywhere (counted vmcnt + lgkmcnt(0) + raw barrier); the halo of chunk c+1 is issued at tap 0
// of chunk c, right before that iteration's weight tile, so for the next NSTW-2 iterations it is YOUNGER than the
// weight tile being waited for and the allowed outstanding count is raised by its (wave-uniform) instruction count.
// ============================================================================
// SC = 1: the variant that accumulates a 1x1 shortcut conv first (GemmDesc::sc_A) -- its own instantiation, so that the
// plain convs keep their register allocation
template <int F16, int TBM, int MINW, int SC = 0>
__global__ __launch_bounds__((TBM / 64) * 2 * 64, MINW) void igemm_halo3x3_kernel(const GemmDesc d,
                                                                                const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];  // [2][HRMAX][32] halo | [NSTW][128][32] weights | dummy
  constexpr int TBN = 128, CK = 32, NSTW = 4;
  constexpr int WM_ = TBM / 64, WN_ = 2, NWAVES = WM_ * WN_;
  constexpr int HRMAX = ((TBM + 2 * 33 + 15) / 16) * 16;        // W <= 32
  constexpr int HGW = (HRMAX / 16 + NWAVES - 1) / NWAVES;       // halo glds instructions per wave per chunk (uniform)
  constexpr int GW = (TBN / 16) / NWAVES >= 1 ? (TBN / 16) / NWAVES : 1;  // weight glds per wave per k-tile
  static_assert((TBN / 16) % NWAVES == 0 || NWAVES > TBN / 16, "weight row groups over the waves");
  constexpr int HBUF = HRMAX * CK, WBUF = TBN * CK;
  op16_t* const hbuf = lds;
  op16_t* const wring = lds + 2 * HBUF;
  op16_t* const dummy = wring + NSTW * WBUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave - wm * WN_;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tile_m = t / d.tiles_n, tile_n = t - tile_m * d.tiles_n;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;
  const int HW = d.rows_per_b, W = d.img_w;
  const int b = m0 / HW, j0 = m0 - b * HW;  // the tile lies inside image b
  const int HR = ((TBM + 2 * (W + 1) + 15) / 16) * 16, HG = HR / 16;

  const int Ktot = 9 * d.Cin;
  const int nchunks = d.Cin / CK;
  const int nkt = 9 * nchunks;

  f32x4 acc[4][4];
  seed_acc<4, 4>(d, acc, m0 + wm * 64, d.M, n0 + wn * 64, lane);  // before any glds: these loads land first (in order)

  const int frow = lane & 15, fchunk = lane >> 4;
  const int w_off = (wn * 64 + frow) * CK + ((fchunk ^ swzk<32>(frow)) * 8);
  const int rsub = lane >> 2, cpos = lane & 3;
  // (the shortcut runs BEFORE the 3x3 loader state is set up: both live at once spilled 39 registers under the cap)
  if constexpr (SC != 0) {
    // ---- 1x1 shortcut conv (GemmDesc::sc_A) accumulated first: per 32-channel chunk the tile's own TBM rows (no halo)
    // and a [128][32] weight tile, two stages (the halo buffers / weight slots 0 and 1), the plain protocol -- wait for
    // the one stage in flight, barrier, issue the next, multiply.  ~0.5 us per chunk against a launch + an fp32
    // round trip of the shortcut's output.
    constexpr int AG = (TBM / 16) / NWAVES;  // row groups of 16 per wave
    const int nch2 = d.sc_Cin / CK;
    const op16_t* asrc[AG];
#pragma unroll
    for (int gi = 0; gi < AG; ++gi) {
      const int h = (wave * AG + gi) * 16 + rsub;
      asrc[gi] = d.sc_A + (long)b * d.sc_bstride + (long)(j0 + h) * d.sc_row_elems + ((cpos ^ swzk<32>(h)) << 3);
    }
    const op16_t* w2src[GW];
    int w2step[GW];
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
      const int g = wave * GW + gi;
      const int row = g * 16 + rsub;
      const bool ok = g < TBN / 16 && n0 + row < d.N;
      w2src[gi] = ok ? d.sc_W + (long)(n0 + row) * d.sc_Cin + ((cpos ^ swzk<32>(row)) << 3) : zero_page + cpos * 8;
      w2step[gi] = ok ? CK : 0;
    }
    auto issue2 = [&](int cc) {
      op16_t* ab = (cc & 1) ? hbuf + HBUF : hbuf;
      op16_t* wb2 = wring + (cc & 1) * WBUF;
#pragma unroll
      for (int gi = 0; gi < AG; ++gi)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[gi] + (long)cc * CK),
                                         (__attribute__((address_space(3))) void*)(ab + (wave * AG + gi) * 16 * CK), 16, 0, 0);
#pragma unroll
      for (int gi = 0; gi < GW; ++gi) {
        const int g = wave * GW + gi;
        op16_t* dst = g < TBN / 16 ? wb2 + g * 16 * CK : dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w2src[gi] + (long)cc * w2step[gi]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    issue2(0);
    for (int cc = 0; cc < nch2; ++cc) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (cc + 1 < nch2) issue2(cc + 1);
      const op16_t* ab = (cc & 1) ? hbuf + HBUF : hbuf;
      const op16_t* wb2 = wring + (cc & 1) * WBUF;
      const int r0 = wm * 64 + frow;  // +16 per sub-tile keeps the swizzle
      const int a_off2 = r0 * CK + ((fchunk ^ swzk<32>(r0)) * 8);
      op16x8 fa[4], fw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fa[k] = *reinterpret_cast<const op16x8*>(ab + a_off2 + k * 16 * CK);
        fw[k] = *reinterpret_cast<const op16x8*>(wb2 + w_off + k * 16 * CK);
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma16<F16>(fw[tn], fa[tm], acc[tn][tm]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read the last stage: the 3x3 prologue may overwrite both
  }
  // ---- loader state ----
  const op16_t* hsrc[HGW];
  int hstep[HGW];
  op16_t* hdst[HGW];
#pragma unroll
  for (int gi = 0; gi < HGW; ++gi) {
    const int g = wave + gi * NWAVES;
    const int h = g * 16 + rsub;
    const int j = j0 - (W + 1) + h;
    const bool ok = g < HG && j >= 0 && j < HW;
    const int gchunk = cpos ^ swzk<32>(h);
    hsrc[gi] = ok ? d.A + (long)b * d.in_bstride + (long)j * d.in_row_elems + gchunk * 8 : zero_page + cpos * 8;
    hstep[gi] = ok ? CK : 0;
    hdst[gi] = g < HG ? hbuf + g * 16 * CK : dummy;
  }
  const op16_t* wsrc[GW];
  int wstep[GW];
#pragma unroll
  for (int gi = 0; gi < GW; ++gi) {
    const int g = wave * GW + gi;
    const int row = g * 16 + rsub;
    const bool live = g < TBN / 16;
    const bool ok = live && n0 + row < d.N;
    const int gchunk = cpos ^ swzk<32>(row);
    wsrc[gi] = ok ? d.W + (long)(n0 + row) * Ktot + gchunk * 8 : zero_page + cpos * 8;
    wstep[gi] = ok ? 1 : 0;
  }
  auto issue_halo = [&](int c) {
    op16_t* base = (c & 1) ? hbuf + HBUF : hbuf;
#pragma unroll
    for (int gi = 0; gi < HGW; ++gi) {
      const op16_t* g = hsrc[gi] + (long)c * hstep[gi];
      op16_t* dst = hdst[gi] == dummy ? dummy : hdst[gi] + (base - hbuf);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  int wi_c = 0, wi_tap = 0;  // (chunk, tap) of the next weight tile to issue
  auto issue_w = [&](int slot) {
    const long off = (long)wi_tap * d.Cin + wi_c * CK;
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
      const int g = wave * GW + gi;
      const op16_t* gp = wsrc[gi] + off * wstep[gi];
      op16_t* dst = g < TBN / 16 ? wring + slot * WBUF + g * 16 * CK : dummy;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    if (++wi_tap == 9) {
      wi_tap = 0;
      ++wi_c;
    }
  };

  // x-border flags of this lane's 4 output rows (row sub-tiles tm): bit tm of xl / xr
  unsigned xl = 0, xr = 0;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int x = (j0 + wm * 64 + tm * 16 + frow) % W;
    if (x == 0) xl |= 1u << tm;
    if (x == W - 1) xr |= 1u << tm;
  }

  issue_halo(0);
#pragma unroll
  for (int s2 = 0; s2 < NSTW - 1; ++s2)
    if (s2 < nkt) issue_w(s2);

  int c = 0, tap = 0;
  for (int i = 0; i < nkt; ++i) {
    const int younger = min(NSTW - 2, nkt - 1 - i);
    const bool halo_younger = (tap == 1 || tap == 2) && c + 1 < nchunks;  // halo(c+1) issued at tap 0, NSTW = 4
    if (halo_younger)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * GW + HGW) : "memory");
    else if (younger >= 2)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * GW) : "memory");
    else if (younger == 1)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if DSN_DBG_MODE != 1
    if (tap == 0 && c + 1 < nchunks) issue_halo(c + 1);
    if (i + NSTW - 1 < nkt) issue_w((i + NSTW - 1) % NSTW);
#endif

    const op16_t* hb = (c & 1) ? hbuf + HBUF : hbuf;
    const op16_t* wb = wring + (i % NSTW) * WBUF;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int h0 = wm * 64 + frow + (W + 1) + dy * W + dx;  // halo row of sub-tile 0; +16 per sub-tile keeps the swizzle
    const int a_off = h0 * CK + ((fchunk ^ swzk<32>(h0)) * 8);
    op16x8 fa[4], fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fa[k] = *reinterpret_cast<const op16x8*>(hb + a_off + k * 16 * CK);
      fw[k] = *reinterpret_cast<const op16x8*>(wb + w_off + k * 16 * CK);
    }
    if (dx != 0) {
      const unsigned mk = dx < 0 ? xl : xr;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((mk >> k) & 1u) fa[k] = op16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#if DSN_DBG_MODE == 2
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(fa[k]), "v"(fw[k]));
#else
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma16<F16>(fw[tn], fa[tm], acc[tn][tm]);
#endif
    if (++tap == 9) {
      tap = 0;
      ++c;
    }
  }
#if DSN_DBG_MODE == 3
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) asm volatile("" ::"v"(acc[a][bb]));
#else
  epilogue_gen<1, F16, 4, 4, 0, LEAN_NO_DIT | LEAN_SEEDED>(d, acc, m0 + wm * 64, d.M, n0 + wn * 64, lane, 0);
  if (d.gnf_out) {
    // ---- GroupNorm finished here (GemmDesc::gnf_out; the epilogue above stored nothing but this tile's slice partials,
    // write-through).  Hand-off as in MI355X_MICROARCH.md, "inter-workgroup visibility", write-through form: every storing
    // wave drains its stores, workgroup barrier, one lane adds to the (image, column tile) counter and polls it until the
    // image's HW / TBM row tiles have added, barrier, then every load of partials is an sc1 load.  The counter only
    // grows (target = next multiple above the value the add returned): nothing to reset between graph replays.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nper = HW / TBM;
    if (tid == 0) {
      unsigned* cnt = d.gnf_sync + (long)b * d.tiles_n + tile_n;
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (old / (unsigned)nper + 1u) * (unsigned)nper;
      int spins = 0;
      while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > (1 << 21) || ((spins & 1023) == 0 && __hip_atomic_load(d.gnf_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
          __hip_atomic_store(d.gnf_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (host: error on the next call)
          break;
        }
      }
    }
    __syncthreads();
    // (mean, rstd) of the groups of this column tile -> LDS (the ring is free).  The combine is gn_apply_kernel's, lane
    // for lane -- TPG lanes per group, lane-local sums in index order, xor tree; its block is 256 threads wide, which
    // fixes TPG -- so both paths normalise with the same bits.
    float* const gm = reinterpret_cast<float*>(lds);
    float* const gr = gm + 64;
    const int C = d.N, nq_all = C >> 2;
    const int G = min(C >> 2, 32), cpg = C / G, qpg = cpg >> 2;
    const int S = HW >> 6;
    int tpg = 1;
    while (tpg < 64 && G * tpg * 2 <= 256) tpg *= 2;
    const int ngt = min(TBN, C - n0) / cpg;  // groups of this column tile
    {
      const int gl = tid / tpg, sub = tid - gl * tpg;
      const bool live = gl < ngt;
      const float* sp = d.gn_stats + ((long)b * S * nq_all + (long)(n0 / cpg + (live ? gl : 0)) * qpg) * 2;
      const int items = S * qpg;
      const float cnt = 256.f;  // 64 rows x 4 channels per partial (whole slices: HW % 64 == 0)
      float wsum = 0.f;
      if (live)
        for (int it = sub; it < items; it += tpg) {
          const int sl = it / qpg, q = it - sl * qpg;
          wsum += cnt * __hip_atomic_load(sp + ((long)sl * nq_all + q) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      for (int o = tpg >> 1; o >= 1; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
      const float ntot = (float)HW * (float)cpg;
      const float mean = wsum / ntot;
      float m2 = 0.f;
      if (live)
        for (int it = sub; it < items; it += tpg) {
          const int sl = it / qpg, q = it - sl * qpg;
          const float* pp = sp + ((long)sl * nq_all + q) * 2;
          const float px = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float py = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float dm = px - mean;
          m2 += py + cnt * dm * dm;
        }
      for (int o = tpg >> 1; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o, 64);
      if (live && sub == 0) {
        gm[gl] = mean;
        gr[gl] = rsqrtf(m2 / ntot + d.gnf_eps);
      }
    }
    __syncthreads();
    const int nqc = (lane >> 4) * 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn * 64 + tn * 16 + nqc;
      if (n >= C) continue;
      const int gl = (n - n0) / cpg;
      const float mean = gm[gl], rstd = gr[gl];
      const f32x4 ga = *reinterpret_cast<const f32x4*>(d.gnf_gamma + n);
      const f32x4 be = *reinterpret_cast<const f32x4*>(d.gnf_beta + n);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wm * 64 + tm * 16 + (lane & 15);
        const f32x4 v = acc[tn][tm] * d.out_scale;
        op16x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = (v[k] - mean) * rstd * ga[k] + be[k];
          h[k] = to_op16(d.gnf_silu ? dsn_silu(t) : t, F16);
        }
        *reinterpret_cast<op16x4*>(d.gnf_out + (long)m * C + n) = h;
      }
    }
  }
#endif
}

// ============================================================================
// Row-panel variant for GEMMs whose M is a small non-multiple of the tile (the DiT's
// M = B*(T+1) = 2112 token rows): the rows are cut into equal PANELS of d.panel_rows (<= MT*16) rows --
// e.g. 8 panels of 264 -- so that panels x column tiles (x split-K) is exactly the CU count and the
// whole GEMM is ONE balanced round.  A workgroup owns one panel x (NWAVES*32) columns; every wave
// holds all MT row sub-tiles x 2 column sub-tiles (acc[2][MT]); the last row sub-tile is partly
// masked.  Staging / ring / swizzle as in igemm2_kernel (BK = 32); row groups are dealt round-robin
// to the waves, so the per-wave glds count (and its vmcnt) differs by one between waves.
// ============================================================================
template <int P, int F16, int WN_, int NST, int TBK, int MT = 17, int EPI = 0, int WM_ = 4>
__global__ __launch_bounds__(WM_ * WN_ * 64, 1) void igemm_panel_kernel(const GemmDesc d,
                                                                         const op16_t* __restrict__ zero_page) {
  // 4 wave rows x WN_ wave columns; the MT row sub-tiles of a panel are dealt MT/4 (+1 for the first MT%4 wave rows):
  // MT = 17: 5/4/4/4 = 272 rows (8 panels of 264 for M = 2112); MT = 9: 3/2/2/2 = 144 rows (16 panels of 132);
  // MT = 7: 2/2/2/1 = 112 rows.  Every wave owns 4 column sub-tiles (64 columns).
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];  // [NST][plane][A rows | W rows (TBN) | pad][TBK]
  // (WM_ = 2: 8 waves with taller wave tiles -- fewer fragment bytes read from LDS per MFMA, 256 registers per lane)
  constexpr int NWAVES = WM_ * WN_;
  constexpr int MTW = (MT + WM_ - 1) / WM_;
  constexpr int TBN = WN_ * 64;
  constexpr int AROWS = MT * 16;
  constexpr int ROWS = AROWS + TBN;
  constexpr int PLANE_ELEMS = ROWS * TBK;
  constexpr int STAGE_ELEMS = P * PLANE_ELEMS;
  constexpr int CPR = TBK / 8;
  constexpr int RPG = 64 / CPR;
  constexpr int GROUPS = ROWS / RPG;
  constexpr int GPW = (GROUPS + NWAVES - 1) / NWAVES;  // max groups per wave
  constexpr int REM = GROUPS % NWAVES;                 // waves < REM carry GPW groups, the rest GPW-1 (REM==0: all GPW)
  constexpr int KS = TBK / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN_, wave_n = wave - wave_m * WN_;
  constexpr int MBASE = MT / WM_, MREM = MT % WM_;
  const int my_mt = MBASE + (wave_m < MREM ? 1 : 0);
  const int my_row0 = 16 * (wave_m * MBASE + min(wave_m, MREM));
  const int my_groups = (REM == 0 || wave < REM) ? GPW : GPW - 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int ntiles = d.tiles_m * d.tiles_n;
  const int z = t / ntiles;
  const int tile = t - z * ntiles;
  const int tile_m = d.m_fast ? tile % d.tiles_m : tile / d.tiles_n;
  const int tile_n = d.m_fast ? tile / d.tiles_m : tile - tile_m * d.tiles_n;
  const int m0 = tile_m * d.panel_rows, n0 = tile_n * TBN;
  const int m_end = min(m0 + d.panel_rows, d.M);

  const int Ktot = d.taps * d.Cin;
  const int kc_per_tap = d.Cin / TBK;
  const int nkt_all = d.taps * kc_per_tap;
  const int kt_begin = (int)((long)nkt_all * z / d.ksplit);
  const int kt_end = (int)((long)nkt_all * (z + 1) / d.ksplit);
  const int nkt = kt_end - kt_begin;

  const int rsub = lane / CPR, cpos = lane % CPR;
  RowLoad rl[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + gi * NWAVES;  // round-robin deal
    const bool is_a = g < AROWS / RPG;
    const int row = (is_a ? g : g - AROWS / RPG) * RPG + rsub;
    const int gchunk = cpos ^ swzk<TBK>(row);
    const int idx = (is_a ? m0 : n0) + row;
    rl[gi] = make_row(d, is_a, idx, g < GROUPS && (is_a ? idx < m_end : idx < d.N), gchunk, Ktot);
  }
  const op16_t* zsrc = zero_page + cpos * 8;
  int itap = kt_begin / kc_per_tap, ikc = kt_begin - itap * kc_per_tap;

  auto issue = [&](int stage) {
    op16_t* sbase = lds + stage * STAGE_ELEMS;
    const long offa = a_tile_off(d, itap, ikc, TBK), offw = w_tile_off(d, itap, ikc, TBK);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (gi < my_groups) {
        const int g = wave + gi * NWAVES;
        const bool ok = (rl[gi].mask >> itap) & 1u;
        const op16_t* g0 = rl[gi].ptr + (rl[gi].is_a ? offa : offw);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const op16_t* gp = ok ? g0 + p * rl[gi].ps : zsrc;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                           (__attribute__((address_space(3))) void*)(sbase + p * PLANE_ELEMS + g * RPG * TBK),
                                           16, 0, 0);
        }
      }
    }
    if (++ikc == kc_per_tap) {
      ikc = 0;
      ++itap;
    }
  };

  f32x4 acc[4][MTW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fchunk = lane >> 4;
  const int fsw = swzk<TBK>(frow);
  const int a_row_off = (my_row0 + frow) * TBK;
  const int w_row_off = (AROWS + wave_n * 64 + frow) * TBK;

#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2)
    if (s2 < nkt) issue(s2);

  if constexpr ((EPI & EPI_STATS) != 0) {
    // residual-stream producer: the accumulators start from x (plain [M][N] fp32), see the epilogue
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm) {
      const int m = m0 + my_row0 + tm * 16 + (lane & 15);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wave_n * 64 + tn * 16 + (lane >> 4) * 4;
        if (tm < my_mt && m < m_end && n < d.N) acc[tn][tm] = *reinterpret_cast<const f32x4*>(d.resid + (long)m * d.N + n);
      }
    }
  }

  // folded LayerNorm: (mean, rstd) of this panel's rows from the producer's per-slice partials, combined in slice
  // order (Chan's parallel formula) -> LDS behind the ring; the first k-tile's barrier publishes them long before
  // the epilogue reads them
  float* const ln_rows = reinterpret_cast<float*>(lds + NST * STAGE_ELEMS);
  if ((EPI & EPI_LNFOLD) && d.ln_stats) {
    for (int r = tid; r < m_end - m0; r += NWAVES * 64) {
      const f32x4* ps = reinterpret_cast<const f32x4*>(d.ln_stats + (long)(m0 + r) * d.ln_np * 2);  // 2 slices each
      float msum = 0.f, m2 = 0.f, q = 0.f;
      // sum of slice means, of slice M2s and of squared slice means in ONE pass (fixed order):
      // M2 = sum M2_p + cnt * (sum mean_p^2 - np * mean^2)
      for (int p2 = 0; p2 < d.ln_np / 2; ++p2) {
        const f32x4 v = ps[p2];
        msum += v[0] + v[2];
        m2 += v[1] + v[3];
        q += v[0] * v[0] + v[2] * v[2];
      }
      const float mean = msum / (float)d.ln_np;
      const float cnt = (float)(d.Cin * d.taps) / (float)d.ln_np;  // columns per slice
      m2 += cnt * fmaxf(q - (float)d.ln_np * mean * mean, 0.f);
      ln_rows[2 * r] = mean;
      ln_rows[2 * r + 1] = rsqrtf(m2 / (float)(d.Cin * d.taps) + d.ln_eps);
    }
  }

  for (int i = 0; i < nkt; ++i) {
    const int younger = min(NST - 2, nkt - 1 - i);
    if (NST >= 4 && younger >= 2) {  // two younger tiles stay in flight
      if (REM == 0 || wave < REM)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * GPW * P) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (GPW - 1) * P) : "memory");
    } else if (NST >= 3 && younger >= 1) {
      if (REM == 0 || wave < REM)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GPW * P) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((GPW - 1) * P) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (i + NST - 1 < nkt) issue((i + NST - 1) % NST);

    const op16_t* base = lds + (i % NST) * STAGE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = ((ks * 4 + fchunk) ^ fsw) * 8;
      op16x8 fw[P][4];
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          fw[p][k] = *reinterpret_cast<const op16x8*>(base + p * PLANE_ELEMS + w_row_off + k * 16 * TBK + coff);
#pragma unroll
      for (int tm = 0; tm < MTW; ++tm) {
        if (tm < my_mt) {
          op16x8 fa[P];
#pragma unroll
          for (int p = 0; p < P; ++p)
            fa[p] = *reinterpret_cast<const op16x8*>(base + p * PLANE_ELEMS + a_row_off + tm * 16 * TBK + coff);
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            if (P == 2) {
              acc[tn][tm] = mfma16<F16>(fw[P - 1][tn], fa[0], acc[tn][tm]);
              acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[P - 1], acc[tn][tm]);
            }
            acc[tn][tm] = mfma16<F16>(fw[0][tn], fa[0], acc[tn][tm]);
          }
        }
      }
    }
  }
  // rows m0 + my_row0 + tm*16 ...; wave rows 1..3 never touch their (unused) 5th sub-tile: mask it by row
  epilogue_gen<P, F16, 4, MTW, EPI, LEAN_NO_NCSN>(d, acc, m0 + my_row0, min(m_end, m0 + my_row0 + my_mt * 16), n0 + wave_n * 64, lane,
                                    z, ((EPI & EPI_LNFOLD) && d.ln_stats) ? ln_rows : nullptr, m0);
}


// ============================================================================
// fp8 (MX) twin of the row-panel kernel (DSN_PREC_FP8: the four DiT layer GEMMs).  Same geometry and ring; a
// k-tile is 128 fp8 per row -- byte for byte the 64-element 16-bit tile -- consumed by ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator (twice the bf16 MFMA rate, half the staged bytes per K).
// The E8M0 block scales (4 bytes per row per k-tile) ride through LDS too: every wave issues one 4-byte-per-lane
// glds per k-tile for the scale dwords of 64 staged rows (waves beyond the staged rows load a zero page), so the
// counted vmcnt stays uniform; lane (r, q) then reads byte q of its row's dword.
// ============================================================================
template <int WM_, int WN_, int NST, int MT, int EPI = EPI_FP8OUT>
__global__ __launch_bounds__(WM_ * WN_ * 64, 1) void igemm_panel_fp8_kernel(const GemmDesc d,
                                                                             const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];  // [NST][A rows | W rows][64 pairs], then [NST][SROWS] u32
  constexpr int TBK = 64;  // byte PAIRS per row per k-tile (128 fp8)
  // WM_ x WN_ waves (8 for the tall panels: the 32-byte fp8 fragments need the 256-register budget of 2 waves per
  // SIMD); the MT row sub-tiles are dealt over the WM_ wave rows
  constexpr int NWAVES = WM_ * WN_;
  constexpr int MTW = (MT + WM_ - 1) / WM_;
  constexpr int TBN = WN_ * 64;
  constexpr int AROWS = MT * 16;
  constexpr int ROWS = AROWS + TBN;
  constexpr int STAGE_ELEMS = ROWS * TBK;
  constexpr int CPR = TBK / 8;
  constexpr int RPG = 64 / CPR;
  constexpr int GROUPS = ROWS / RPG;
  constexpr int GPW = (GROUPS + NWAVES - 1) / NWAVES;
  constexpr int REM = GROUPS % NWAVES;
  constexpr int SG = (ROWS + 63) / 64;             // 64-row scale groups that hold staged rows
  constexpr int SGW = (SG + NWAVES - 1) / NWAVES;  // scale groups per wave (uniform: tail slots load a zero page)
  constexpr int SROWS = (SG + 1) * 64;             // scale dwords per stage; slots >= SG land in the spare group
  constexpr bool PF = MTW < 8;                     // A fragments one row sub-tile ahead of their MFMAs
  unsigned* const slds = reinterpret_cast<unsigned*>(lds + NST * STAGE_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN_, wave_n = wave - wave_m * WN_;
  constexpr int MBASE = MT / WM_, MREM = MT % WM_;
  const int my_mt = MBASE + (wave_m < MREM ? 1 : 0);
  const int my_row0 = 16 * (wave_m * MBASE + min(wave_m, MREM));
  const int my_groups = (REM == 0 || wave < REM) ? GPW : GPW - 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int ntiles = d.tiles_m * d.tiles_n;
  const int z = t / ntiles;
  const int tile = t - z * ntiles;
  const int tile_m = d.m_fast ? tile % d.tiles_m : tile / d.tiles_n;
  const int tile_n = d.m_fast ? tile / d.tiles_m : tile - tile_m * d.tiles_n;
  const int m0 = tile_m * d.panel_rows, n0 = tile_n * TBN;
  const int m_end = min(m0 + d.panel_rows, d.M);

  const int Ktot = d.Cin;             // byte pairs per row (taps == 1)
  const int nkt_all = d.Cin / TBK;
  const int kt_begin = (int)((long)nkt_all * z / d.ksplit);
  const int kt_end = (int)((long)nkt_all * (z + 1) / d.ksplit);
  const int nkt = kt_end - kt_begin;

  const int rsub = lane / CPR, cpos = lane % CPR;
  // this lane's 16-byte slot of each staged row at k = 0, as a 32-bit element offset from the (wave-uniform) operand
  // base.  Rows outside M / N are CLAMPED to the last valid row instead of routed to a zero page: they only feed
  // accumulators of rows / columns the epilogue never stores, and a uniform base + 32-bit offset costs half the
  // registers of a pointer per group (the 17-sub-tile tile needs them)
  unsigned roff[GPW];  // (unsigned: base + zero-extended offset is the scalar-base addressing form of the load)
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + gi * NWAVES;
    const bool is_a = g < AROWS / RPG;
    const int row = (is_a ? g : g - AROWS / RPG) * RPG + rsub;
    const int gchunk = cpos ^ swzk<TBK>(row);
    const int idx = min((is_a ? m0 : n0) + row, (is_a ? m_end : d.N) - 1);
    roff[gi] = (unsigned)(idx * Ktot + gchunk * 8) * 2u;  // BYTES (no shift between the zero-extension and the add)
  }
  // scale loader: slot j of this wave = rows (wave + j*NWAVES)*64 + lane of the staged panel (clamped the same way;
  // slots past the staged rows re-read row 0 of the weight tile and land in the spare group)
  unsigned soff[SGW];
  bool s_a[SGW];
#pragma unroll
  for (int j = 0; j < SGW; ++j) {
    const int srow = (wave + j * NWAVES) * 64 + lane;
    const bool s_is_a = srow < AROWS;
    s_a[j] = s_is_a;
    const int sidx = s_is_a ? min(m0 + srow, m_end - 1) : min(n0 + min(srow - AROWS, TBN - 1), d.N - 1);
    soff[j] = (unsigned)(sidx * d.mx_kblocks);
  }
  int kt_abs = kt_begin;

  auto issue = [&](int stage) {
    op16_t* sbase = lds + stage * STAGE_ELEMS;
    const unsigned off = (unsigned)(kt_abs * TBK) * 2u;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (gi < my_groups) {
        const int g = wave + gi * NWAVES;
        const char* gp = reinterpret_cast<const char*>(g < AROWS / RPG ? d.A : d.W) + (off + roff[gi]);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(sbase + g * RPG * TBK), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < SGW; ++j) {
      const unsigned char* sp = (s_a[j] ? d.a_scale : d.w_scale) + (soff[j] + (unsigned)(4 * kt_abs));
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)sp,
          (__attribute__((address_space(3))) void*)(slds + stage * SROWS + min(wave + j * NWAVES, SG) * 64), 4, 0, 0);
    }
    ++kt_abs;
  };

  f32x4 acc[4][MTW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fchunk = lane >> 4;
  const int fsw = swzk<TBK>(frow);
  const int a_row_off = (my_row0 + frow) * TBK;
  const int w_row_off = (AROWS + wave_n * 64 + frow) * TBK;
  const int c_lo = (fchunk ^ fsw) * 8, c_hi = ((4 + fchunk) ^ fsw) * 8;

#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2)
    if (s2 < nkt) issue(s2);

  if constexpr ((EPI & EPI_STATS) != 0) {
    // residual-stream producer (folded ff_norm, fp8 mode): the accumulators start from x (plain [M][N] fp32); the MFMA
    // scales apply to the products only
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm) {
      const int m = m0 + my_row0 + tm * 16 + (lane & 15);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wave_n * 64 + tn * 16 + (lane >> 4) * 4;
        if (tm < my_mt && m < m_end && n < d.N) acc[tn][tm] = *reinterpret_cast<const f32x4*>(d.resid + (long)m * d.N + n);
      }
    }
  }
  // folded LayerNorm: (mean, rstd) of the panel's rows from the producer's per-64-column partials (as in
  // igemm_panel_kernel; the row length is the fp8 K = 2 * Cin) -> LDS behind the scale ring
  float* const ln_rows = reinterpret_cast<float*>(slds + NST * SROWS);
  if ((EPI & EPI_LNFOLD) && d.ln_stats) {
    const float kcols = (float)(2 * d.Cin);
    for (int r = tid; r < m_end - m0; r += NWAVES * 64) {
      const f32x4* ps = reinterpret_cast<const f32x4*>(d.ln_stats + (long)(m0 + r) * d.ln_np * 2);
      float msum = 0.f, m2 = 0.f, q = 0.f;
      for (int p2 = 0; p2 < d.ln_np / 2; ++p2) {
        const f32x4 v = ps[p2];
        msum += v[0] + v[2];
        m2 += v[1] + v[3];
        q += v[0] * v[0] + v[2] * v[2];
      }
      const float mean = msum / (float)d.ln_np;
      m2 += (kcols / (float)d.ln_np) * fmaxf(q - (float)d.ln_np * mean * mean, 0.f);
      ln_rows[2 * r] = mean;
      ln_rows[2 * r + 1] = rsqrtf(m2 / kcols + d.ln_eps);
    }
  }

  for (int i = 0; i < nkt; ++i) {
    const int younger = min(NST - 2, nkt - 1 - i);
    if (NST >= 3 && younger >= 1) {
      if (REM == 0 || wave < REM)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GPW + SGW) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GPW - 1 + SGW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (i + NST - 1 < nkt) issue((i + NST - 1) % NST);

    const op16_t* base = lds + (i % NST) * STAGE_ELEMS;
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(slds + (i % NST) * SROWS) + fchunk;
    op16x8 fw0[4], fw1[4];
    int sw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fw0[k] = *reinterpret_cast<const op16x8*>(base + w_row_off + k * 16 * TBK + c_lo);
      fw1[k] = *reinterpret_cast<const op16x8*>(base + w_row_off + k * 16 * TBK + c_hi);
      sw[k] = sb[4 * (AROWS + wave_n * 64 + k * 16 + frow)];
    }
    // activation fragments one row sub-tile ahead of the MFMAs that use them; the scheduling fence per sub-tile keeps
    // hipcc from hoisting every fragment read to the top (9 x 8 registers: spills at MT = 17)
    op16x8 fa0 = *reinterpret_cast<const op16x8*>(base + a_row_off + c_lo);
    op16x8 fa1 = *reinterpret_cast<const op16x8*>(base + a_row_off + c_hi);
    int sa = sb[4 * (my_row0 + frow)];
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm) {
      op16x8 na0 = fa0, na1 = fa1;
      int nsa = sa;
      if (PF && tm + 1 < MTW) {
        na0 = *reinterpret_cast<const op16x8*>(base + a_row_off + (tm + 1) * 16 * TBK + c_lo);
        na1 = *reinterpret_cast<const op16x8*>(base + a_row_off + (tm + 1) * 16 * TBK + c_hi);
        nsa = sb[4 * (my_row0 + (tm + 1) * 16 + frow)];
      }
      // (tall wave tiles run branch-free: a wave row's missing last sub-tile reads the rows that follow in LDS and its
      // accumulators are never stored -- with the uniform branch hipcc spills fragments and accumulators in the loop)
      if (!PF || tm < my_mt) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tn][tm] = mfma_mx8(fw0[tn], fw1[tn], fa0, fa1, acc[tn][tm], sw[tn], sa);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!PF && tm + 1 < MTW) {  // tall wave tiles: no registers for a second fragment pair
        const int ao = a_row_off + (tm + 1) * 16 * TBK;
        na0 = *reinterpret_cast<const op16x8*>(base + ao + c_lo);
        na1 = *reinterpret_cast<const op16x8*>(base + ao + c_hi);
        nsa = sb[4 * (my_row0 + (tm + 1) * 16 + frow)];
      }
      fa0 = na0;
      fa1 = na1;
      sa = nsa;
    }
  }
  epilogue_gen<1, 1, 4, MTW, EPI, LEAN_NO_NCSN>(d, acc, m0 + my_row0, min(m_end, m0 + my_row0 + my_mt * 16), n0 + wave_n * 64, lane, z,
                                  ((EPI & EPI_LNFOLD) && d.ln_stats) ? ln_rows : nullptr, m0);
}


// ============================================================================
// Skinny GEMM for M <= 48 rows (one mixture, or a handful: config C1 is 17 token rows): there the path is bound by
// streaming the weights once and by launch latency, not by MFMA work, and a tile kernel leaves most CUs idle
// (QKV at 256-column tiles: 12 workgroups).  A 4-wave workgroup owns 32 output columns (x split-K): its weight
// rows go straight from global memory into MFMA fragments (each element is used by this wave only: no LDS, no
// barrier), the few activation rows come from L2 the same way; 8 k-steps of loads are in flight per wave.
// N / 32 x ksplit workgroups: QKV 96, FF-in 256, to_out / FF-out 32 x 8 = 256.
// ============================================================================
template <int F16, int MT>
__global__ __launch_bounds__(256) void igemm_skinny_kernel(const GemmDesc d) {
  // 4 waves per workgroup split its K range (more loads in flight per CU: the kernel is one HBM round trip long);
  // their partial tiles are summed through LDS in wave order (deterministic) by wave 0, which runs the epilogue
  constexpr int KW = 4;
  constexpr int U = MT <= 4 ? DSN_SKINNY_U : 2;  // k-steps of 32 per load batch (registers: U * (MT + 2) fragments)
  __shared__ f32x4 red[KW - 1][2 * MT][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int tiles_n = (d.N + 31) >> 5;
  const int z = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x - z * tiles_n) * 32;
  const int K = d.Cin;
  const int nk = K >> 5;
  const int kb0 = (int)((long)nk * z / d.ksplit), ke0 = (int)((long)nk * (z + 1) / d.ksplit);
  const int kb = kb0 + (int)((long)(ke0 - kb0) * wave / KW), ke = kb0 + (int)((long)(ke0 - kb0) * (wave + 1) / KW);
  const op16_t* wp[2];
  const op16_t* ap[MT];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) wp[tn] = d.W + (long)min(n0 + tn * 16 + r, d.N - 1) * K + q * 8;
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) ap[tm] = d.A + (long)min(tm * 16 + r, d.M - 1) * d.in_row_elems + q * 8;
  f32x4 acc[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += U) {
    op16x8 fw[U][2], fa[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = min(k0 + u, ke - 1);  // tail steps re-read the last tile and are skipped below
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) fw[u][tn] = *reinterpret_cast<const op16x8*>(wp[tn] + ks * 32);
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) fa[u][tm] = *reinterpret_cast<const op16x8*>(ap[tm] + ks * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u < ke) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int tm = 0; tm < MT; ++tm) acc[tn][tm] = mfma16<F16>(fw[u][tn], fa[u][tm], acc[tn][tm]);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) red[wave - 1][tn * MT + tm][lane] = acc[tn][tm];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < KW - 1; ++w)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) acc[tn][tm] += red[w][tn * MT + tm][lane];
  epilogue_gen<1, F16, 2, MT>(d, acc, 0, d.M, n0, lane, z);
}

const op16_t* zero_page() {
  static op16_t* zp[64] = {};
  op16_t*& z = zp[dsn_current_device()];
  if (!z) {
    if (hipMalloc((void**)&z, 4096) != hipSuccess) return nullptr;
    (void)hipMemset(z, 0, 4096);
  }
  return z;
}

template <int P, int F16, int TBM, int TBN, int NST, int TBK, int LEAN = 0, int WTN = 64>
hipError_t launch_cfg(GemmDesc d, const op16_t* zp, hipStream_t stream) {
  d.tiles_m = cdiv(d.M, TBM);
  d.tiles_n = cdiv(d.N, TBN);
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm2_kernel<P, F16, TBM, TBN, NST, TBK, LEAN, WTN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = d.tiles_m * d.tiles_n * d.ksplit;
  const size_t smem = (size_t)NST * P * (TBM + TBN) * TBK * sizeof(op16_t);
  hipLaunchKernelGGL((igemm2_kernel<P, F16, TBM, TBN, NST, TBK, LEAN, WTN>), dim3(grid), dim3((TBM / 64) * (TBN / WTN) * 64),
                     smem, stream, d, zp);
  return hipGetLastError();
}

}  // namespace

hipError_t igemm2_launch_cfg(const GemmDesc& din, int pl, int bm, int bn, int nst, int bk, hipStream_t stream) {
  const int planes = PL_COUNT(pl), f16 = PL_F16(pl);
  GemmDesc d = din;

  if (d.ksplit < 1) d.ksplit = 1;
  if (d.Cin % bk != 0 || d.M <= 0 || d.N <= 0) return hipErrorInvalidValue;
  if (d.swiglu && (d.N % 32 != 0)) return hipErrorInvalidValue;
  if (d.ksplit > 1 && (!d.out_f32 || d.swiglu)) return hipErrorInvalidValue;
  const op16_t* zp = zero_page();
  if (!zp) return hipErrorOutOfMemory;
  // conv-stack descriptors (Oobleck: bias / activation / residual / fp32 + plane outputs only) take the lean epilogue
  const bool lean = !d.rope_cos && d.qkv_D <= 0 && !d.swiglu && !d.gn_stats && !d.bbias && d.f32_op != DSN_F32_TANH &&
                    d.ksplit <= 1;
#define CFGL(P_, BM_, BN_, NS_, BK_)                                                                   \
  if (lean && planes == P_ && bm == BM_ && bn == BN_ && nst == NS_ && bk == BK_)                       \
    return f16 ? launch_cfg<P_, 1, BM_, BN_, NS_, BK_, LEAN_NO_DIT | LEAN_NO_NCSN | LEAN_SEEDED>(d, zp, stream)      \
               : launch_cfg<P_, 0, BM_, BN_, NS_, BK_, LEAN_NO_DIT | LEAN_NO_NCSN | LEAN_SEEDED>(d, zp, stream);
  CFGL(1, 256, 256, 2, 64) CFGL(1, 256, 256, 3, 32) CFGL(2, 256, 256, 2, 32) CFGL(1, 256, 128, 3, 64)
  CFGL(1, 256, 128, 3, 32) CFGL(1, 256, 128, 2, 64) CFGL(1, 128, 128, 3, 32) CFGL(1, 128, 128, 2, 64)
#undef CFGL
#define CFG(P_, BM_, BN_, NS_, BK_)                                                   \
  if (planes == P_ && bm == BM_ && bn == BN_ && nst == NS_ && bk == BK_)              \
    return f16 ? launch_cfg<P_, 1, BM_, BN_, NS_, BK_>(d, zp, stream)                 \
               : launch_cfg<P_, 0, BM_, BN_, NS_, BK_>(d, zp, stream);
  if (d.sc_A || (d.gnf_out && !(planes == 1 && bm == 128 && bn == 64 && nst == 3 && bk == 64 && igemm2_gnfin_ok(d, pl) &&
                                d.gnf_gamma && d.gnf_beta && !d.out_f32 && !d.out_planes)))
    return hipErrorInvalidValue;
  if (planes == 1 && bm == 128 && bn == 64 && nst == 3 && bk == 64)   // 4 waves of 64 x 32 (NCSN++ level 2)
    return f16 ? launch_cfg<1, 1, 128, 64, 3, 64, 0, 32>(d, zp, stream) : launch_cfg<1, 0, 128, 64, 3, 64, 0, 32>(d, zp, stream);
  // split (2-plane) modes: 48 MFMAs per wave per 32-deep k-tile already amortise the barrier
  CFG(2, 128, 128, 2, 32) CFG(2, 256, 128, 2, 32) CFG(2, 128, 256, 2, 32) CFG(2, 256, 256, 2, 32)
  CFG(2, 256, 128, 3, 32)
  // single-plane modes
  CFG(1, 128, 128, 3, 32) CFG(1, 256, 128, 3, 32) CFG(1, 256, 256, 3, 32)
  CFG(1, 128, 128, 2, 64) CFG(1, 128, 128, 3, 64) CFG(1, 256, 128, 2, 64) CFG(1, 128, 256, 2, 64)
  CFG(1, 256, 256, 2, 64) CFG(1, 256, 128, 3, 64) CFG(1, 128, 256, 3, 64)
#undef CFG
  return hipErrorInvalidValue;
}

// halo-resident 3x3 conv: eligibility + launch (hipErrorNotSupported = not eligible, caller falls back)
// MINW = 4: registers capped at 128 so that two 8-wave workgroups share a CU (123 VGPRs, no spills since the GroupNorm
// partials are taken in one pass); MINW = 1 for the 4-wave 128-row variant.
template <int F16, int TBM, int MINW, int SC = 0>
static hipError_t launch_halo_t(GemmDesc d, const op16_t* zp, hipStream_t stream) {
  d.tiles_m = d.M / TBM;
  d.tiles_n = cdiv(d.N, 128);
  constexpr int HRMAX = ((TBM + 2 * 33 + 15) / 16) * 16;
  const size_t smem = (size_t)(2 * HRMAX * 32 + 4 * 128 * 32 + 16 * 32) * sizeof(op16_t);
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_halo3x3_kernel<F16, TBM, MINW, SC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((igemm_halo3x3_kernel<F16, TBM, MINW, SC>), dim3(d.tiles_m * d.tiles_n), dim3((TBM / 64) * 2 * 64), smem,
                     stream, d, zp);
  return hipGetLastError();
}
// which halo variant runs `d`: 256 (8 waves, 256-row tiles), 128 (4 waves), 0 = not eligible
static int halo_variant(const GemmDesc& d, int pl) {
  if (PL_COUNT(pl) != 1 || d.img_w <= 0 || d.img_w > 32 || d.taps != 9 || d.Cin % 32 != 0 || d.in_stride != 1 ||
      d.ksplit > 1 || d.swiglu || d.rows_per_b % 256 != 0 || d.rows_per_b != d.img_w * d.img_h || d.M % 256 != 0 ||
      d.in_pad != 0)
    return 0;
  return (long)(d.M / 256) * cdiv(d.N, 128) >= 2 * 256 ? 256 : 128;
}
template <int F16, int TBM, int MINW>
static int halo_resident_blocks() {
  constexpr int HRMAX = ((TBM + 2 * 33 + 15) / 16) * 16;
  const size_t smem = (size_t)(2 * HRMAX * 32 + 4 * 128 * 32 + 16 * 32) * sizeof(op16_t);
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_halo3x3_kernel<F16, TBM, MINW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int per_cu = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, igemm_halo3x3_kernel<F16, TBM, MINW>, (TBM / 64) * 2 * 64,
                                                   smem) != hipSuccess ||
      hipGetDeviceProperties(&prop, dsn_current_device()) != hipSuccess)
    return 0;
  return per_cu * prop.multiProcessorCount;
}
// GroupNorm finished inside igemm2's 128 x 64 tile (128-pixel images: one workgroup holds the image).  The caller
// forces that tile (cfg_bm/bn/nst/bk = 128, 64, 3, 64).
bool igemm2_gnfin_ok(const GemmDesc& d, int pl) {
  if (getenv("DSN_NO_GN_FIN") != nullptr || getenv("DSN_NO_GN_FIN_L2") != nullptr || PL_COUNT(pl) != 1 || !d.gn_stats || d.rows_per_b != 128 || d.M % 128 != 0 ||
      d.N % 64 != 0 || d.N > 1024 || d.Cin % 64 != 0 || d.ksplit > 1 || d.resid || d.out_scale != 1.f || d.swiglu ||
      d.rope_cos || d.qkv_D > 0)
    return false;
  const int G = std::min(d.N / 4, 32), cpg = d.N / G;
  return cpg % 4 == 0 && 64 % cpg == 0;
}
bool igemm_halo3x3_eligible(const GemmDesc& d, int pl) {
  return getenv("DSN_NO_HALO") == nullptr && halo_variant(d, pl) != 0;
}
bool igemm_halo3x3_gnfin_ok(const GemmDesc& d, int pl) {
  const bool off = getenv("DSN_NO_GN_FIN") != nullptr || getenv("DSN_NO_HALO") != nullptr;  // (per call: tests flip it)
  const int v = halo_variant(d, pl);
  if (off || !v || !d.gn_stats || d.N % 32 != 0 || d.N > 1024 || d.rows_per_b % v != 0 || d.out_scale != 1.f) return false;
  const int G = std::min(d.N / 4, 32), cpg = d.N / G;
  if (cpg % 4 != 0 || 128 % cpg != 0) return false;  // groups made of whole quads, never across a column tile
  static int cap[2][2] = {{-1, -1}, {-1, -1}};
  int& c = cap[PL_F16(pl) ? 1 : 0][v == 256 ? 1 : 0];
  if (c < 0)
    c = v == 256 ? (PL_F16(pl) ? halo_resident_blocks<1, 256, 4>() : halo_resident_blocks<0, 256, 4>())
                 : (PL_F16(pl) ? halo_resident_blocks<1, 128, 1>() : halo_resident_blocks<0, 128, 1>());
  return (long)(d.M / v) * cdiv(d.N, 128) <= c;  // every workgroup resident at once: they wait for each other
}
hipError_t igemm_halo3x3_launch(const GemmDesc& din, int pl, hipStream_t stream) {
  const int planes = PL_COUNT(pl), f16 = PL_F16(pl);
  const GemmDesc& d = din;
  if (d.gnf_out && (!igemm_halo3x3_gnfin_ok(d, pl) || !d.gnf_gamma || !d.gnf_beta || !d.gnf_sync || !d.gnf_err ||
                    d.out_f32 || d.out_planes))
    return hipErrorInvalidValue;
  if (d.sc_A && (!d.sc_W || d.sc_Cin < 32 || d.sc_Cin % 32 != 0 || d.sc_row_elems < d.sc_Cin)) return hipErrorInvalidValue;
  // 128-row images (NCSN++ level 2, 128 workgroups of 4 waves) measured slower than igemm2's 128 x 128 x BK 64 tiles
  // (62 vs 55 us): not routed here
  if (planes != 1 || d.img_w <= 0 || d.img_w > 32 || d.taps != 9 || d.Cin % 32 != 0 || d.in_stride != 1 ||
      d.ksplit > 1 || d.swiglu || d.rows_per_b % 256 != 0 || d.rows_per_b != d.img_w * d.img_h || d.M % 256 != 0 ||
      d.in_pad != 0)
    return hipErrorNotSupported;
  const op16_t* zp = zero_page();
  if (!zp) return hipErrorOutOfMemory;
  const long wgs = (long)(d.M / 256) * cdiv(d.N, 128);
  if (d.sc_A) {
    if (wgs >= 2 * 256) return f16 ? launch_halo_t<1, 256, 4, 1>(d, zp, stream) : launch_halo_t<0, 256, 4, 1>(d, zp, stream);
    return f16 ? launch_halo_t<1, 128, 1, 1>(d, zp, stream) : launch_halo_t<0, 128, 1, 1>(d, zp, stream);
  }
  if (wgs >= 2 * 256)  // MI355X: 256 CUs
    return f16 ? launch_halo_t<1, 256, 4>(d, zp, stream) : launch_halo_t<0, 256, 4>(d, zp, stream);
  // fewer than two 256-row workgroups per CU (NCSN++ level 1: 256): 128-row tiles of 4 waves, two of them per CU
  // (whole score call 5.48 -> 5.38 ms)
  return f16 ? launch_halo_t<1, 128, 1>(d, zp, stream) : launch_halo_t<0, 128, 1>(d, zp, stream);
}

// Split-K for conv GEMMs whose output tiles fill only a fraction of the chip (single mixtures: a 3x3 conv of the NCSN++
// level 0 is 8 tiles): the GEMM writes raw partial sums to `nslab` slabs (GemmDesc::ksplit), this kernel sums them in
// slab order into the accumulator layout -- one wave per 64 x 64 tile, bias / per-item bias / residual fetched with the
// slabs (seed_acc) -- and runs the ordinary epilogue (scale, GroupNorm partials, fp32 / plane outputs): one code path
// for both, deterministic.  (At the C2 batch the pass costs 18-21 us and loses against the unsplit GEMM; it is used
// for small grids only.)
namespace {
template <int F16>
__global__ __launch_bounds__(64) void igemm_slab_epilogue_kernel(GemmDesc d, const float* __restrict__ slabs, int nslab,
                                                                 long slab_stride) {
  const int lane = threadIdx.x & 63;
  const int tiles_n = (d.N + 63) >> 6;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int mw0 = tile_m * 64, nw0 = tile_n * 64;
  const int nq = (lane >> 4) * 4;
  f32x4 acc[4][4];
  long offs[4];
  bool rowok[4];
  seed_acc<4, 4>(d, acc, mw0, d.M, nw0, lane);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = mw0 + tm * 16 + (lane & 15);
    const bool mok = m < d.M;
    const int b = mok ? m / d.rows_per_b : 0;
    const int j = mok ? m - b * d.rows_per_b : 0;
    const long row_rel = (long)j * d.out_row_elems + d.out_off;
    offs[tm] = (long)b * d.out_bstride + row_rel;
    rowok[tm] = mok && row_rel + nw0 >= 0 && row_rel + nw0 + 64 <= d.out_limit;
  }
  for (int z = 0; z < nslab; ++z) {
    f32x4 part[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = nw0 + tn * 16 + nq;
        part[tn][tm] = (rowok[tm] && n < d.N) ? *reinterpret_cast<const f32x4*>(slabs + z * slab_stride + offs[tm] + n)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tn][tm] += part[tn][tm];
  }
  epilogue_gen<1, F16, 4, 4, 0, LEAN_NO_DIT | LEAN_SEEDED>(d, acc, mw0, d.M, nw0, lane, 0);
}
}  // namespace

hipError_t igemm_slab_epilogue_launch(const GemmDesc& din, int pl, const float* slabs, int nslab, long slab_stride,
                                      hipStream_t stream) {
  if (PL_COUNT(pl) != 1 || din.swiglu || din.rope_cos || nslab < 1 || din.N % 64 != 0) return hipErrorInvalidValue;
  GemmDesc d = din;
  d.ksplit = 1;
  const int grid = cdiv(d.M, 64) * cdiv(d.N, 64);
  if (PL_F16(pl)) hipLaunchKernelGGL(igemm_slab_epilogue_kernel<1>, dim3(grid), dim3(64), 0, stream, d, slabs, nslab, slab_stride);
  else hipLaunchKernelGGL(igemm_slab_epilogue_kernel<0>, dim3(grid), dim3(64), 0, stream, d, slabs, nslab, slab_stride);
  return hipGetLastError();
}

hipError_t igemm2_launch(const GemmDesc& d, int pl, hipStream_t stream) {
  const int planes = PL_COUNT(pl);
  // Tile choice from the measured sweeps (scripts/gemm_bench.py; profiles/r01_gemm_sweep_*.log).
  // The kernel is L2->LDS bound and, at the DiT's M ~ 2k rows, wave-quantisation bound: take the
  // biggest tile whose grid still fills 256 CUs.
  if (d.cfg_bm > 0) return igemm2_launch_cfg(d, pl, d.cfg_bm, d.cfg_bn, d.cfg_nst, d.cfg_bk, stream);
  if (d.img_w > 0 && getenv("DSN_NO_HALO") == nullptr) {  // 3x3 convs: halo-resident kernel where it applies
    const hipError_t e = igemm_halo3x3_launch(d, pl, stream);
    if (e != hipErrorNotSupported) return e;
  }
  if (d.gnf_out || d.sc_A) return hipErrorInvalidValue;  // (the halo kernel, or the forced 128 x 64 tile above, take these)
  int bm = 128, bn = 128, nst = planes == 2 ? 2 : 3, bk = 32;
  const bool k64 = planes == 1 && d.Cin % 64 == 0;
  auto tiles = [&](int tm, int tn) { return (long)cdiv(d.M, tm) * cdiv(d.N, tn) * std::max(d.ksplit, 1); };
  if (planes == 2) {
    // split (hi,lo) operands: 48 MFMAs per wave per 32-deep k-tile
    if (d.ksplit <= 1 && d.taps * d.Cin <= 256) return igemm_launch(d, pl, stream);  // epilogue-bound 1x1 convs
    if (d.M >= 8192) {
      if (d.N >= 256 && tiles(256, 256) >= 256) bm = bn = 256;
    } else if (d.M >= 4096 && d.N >= 4096) {
      bm = 256;
      nst = 3;
    }
  } else if (d.M >= 8192) {                  // large-M conv stacks: the biggest tile that still fills 256 CUs
    if (d.N >= 256 && tiles(256, 256) >= 256) {
      bm = bn = 256;
      if (k64) { nst = 2; bk = 64; }
    } else if (d.N < 256 || tiles(256, 128) >= 256) {   // (NCSN++ level 1: M = 32768, N = 256)
      bm = 256;
      if (k64 && d.N >= 256) { nst = 2; bk = 64; }
    } else if (k64) {                                    // (NCSN++ level 2: M = 8192)
      bk = 64;
      // 128 x 128 tiles would leave half the chip idle (64 x 2 workgroups): 128 x 64 tiles of four 64 x 32 waves
      static const bool no_n64 = getenv("DSN_NO_N64_TILE") != nullptr;
      if (!no_n64 && d.ksplit <= 1 && tiles(128, 128) <= 160 && d.N % 64 == 0) bn = 64;
    }
  } else if (d.M >= 4096 && d.N >= 4096) {   // ConvTranspose phase GEMMs
    bm = bn = 256;
    if (k64) { nst = 2; bk = 64; } else { bn = 128; }
  } else if (d.N >= 4096) {                  // DiT FF-in
    bm = 256;
  } else if (d.N >= 2048 && k64) {           // DiT QKV
    bm = 256;
    bk = 64;
  } else if (d.ksplit > 1 && k64) {          // DiT residual-stream GEMMs, split-K
    bm = 256;
    nst = 2;
    bk = 64;
  }
  return igemm2_launch_cfg(d, pl, bm, bn, nst, bk, stream);
}

// Row-panel launcher: d.panel_rows rows per workgroup (<= 272), bn in {128, 256}.
template <int P, int F16, int WN_, int NST, int TBK, int MT = 17, int EPI = 0, int WM_ = 4>
static hipError_t launch_panel_t(GemmDesc d, const op16_t* zp, hipStream_t stream) {
  constexpr int TBN = WN_ * 64;
  d.tiles_m = cdiv(d.M, d.panel_rows);
  d.tiles_n = cdiv(d.N, TBN);
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_panel_kernel<P, F16, WN_, NST, TBK, MT, EPI, WM_>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = d.tiles_m * d.tiles_n * d.ksplit;
  const size_t smem = (size_t)NST * P * (MT * 16 + TBN) * TBK * sizeof(op16_t) + (d.ln_stats ? MT * 16 * 2 * sizeof(float) : 0);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((igemm_panel_kernel<P, F16, WN_, NST, TBK, MT, EPI, WM_>), dim3(grid), dim3(WM_ * WN_ * 64), smem, stream,
                     d, zp);
  return hipGetLastError();
}


template <int WM_, int WN_, int NST, int MT, int EPI = EPI_FP8OUT>
static hipError_t launch_panel_fp8_t(GemmDesc d, const op16_t* zp, hipStream_t stream) {
  constexpr int TBN = WN_ * 64;
  d.tiles_m = cdiv(d.M, d.panel_rows);
  d.tiles_n = cdiv(d.N, TBN);
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_panel_fp8_kernel<WM_, WN_, NST, MT, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = d.tiles_m * d.tiles_n * d.ksplit;
  const size_t smem = (size_t)NST * ((MT * 16 + TBN) * 64 * sizeof(op16_t) + ((MT * 16 + TBN + 63) / 64 + 1) * 64 * sizeof(unsigned)) +
                      ((EPI & EPI_LNFOLD) ? MT * 16 * 2 * sizeof(float) : 0);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((igemm_panel_fp8_kernel<WM_, WN_, NST, MT, EPI>), dim3(grid), dim3(WM_ * WN_ * 64), smem, stream, d, zp);
  return hipGetLastError();
}

hipError_t igemm_panel_fp8_launch(const GemmDesc& din, int bn, hipStream_t stream) {
  GemmDesc d = din;
  if (d.ksplit < 1) d.ksplit = 1;
  // plain row-major GEMM only: one "batch item" of M rows, one tap, K = 2 * Cin fp8 per row, whole 128-wide k-tiles
  if (!d.a_scale || !d.w_scale || d.taps != 1 || d.in_stride != 1 || d.in_pad != 0 || d.rows_per_b != d.M ||
      d.img_w > 0 || d.Cin % 64 != 0 || d.mx_kblocks != d.Cin / 16 || d.M <= 0 || d.N <= 0 || d.panel_rows <= 0 ||
      d.panel_rows > 17 * 16 || (long)std::max(d.M, d.N) * d.Cin * 2 >= (1L << 32))
    return hipErrorInvalidValue;
  if (d.swiglu && (d.N % 64 != 0)) return hipErrorInvalidValue;
  if (d.ksplit > 1 && (!d.out_f32 || d.swiglu)) return hipErrorInvalidValue;
  // folded ff_norm: the producer (statistics + raw fp8 x') and the SwiGLU consumer, as in igemm_panel_launch
  if (d.stat_out && (d.N % 64 != 0 || d.stat_np != d.N / 64 || d.swiglu || d.ksplit > 1 || !d.resid || !d.out_f32 ||
                     !d.out_fp8 || !d.out_fp8_scale))
    return hipErrorInvalidValue;
  if (d.ln_stats && (!d.swiglu || !d.ln_colsum || d.ln_np <= 0 || (d.ln_np & 1) || !d.out_fp8 || !d.out_fp8_scale))
    return hipErrorInvalidValue;
  const op16_t* zp = zero_page();
  if (!zp) return hipErrorOutOfMemory;
#define FCFGE(MT_, WM_, W_, NS_, E_) \
  if (d.panel_rows <= MT_ * 16 && bn == W_ * 64) return launch_panel_fp8_t<WM_, W_, NS_, MT_, E_>(d, zp, stream);
#define FCFG(MT_, WM_, W_, NS_) FCFGE(MT_, WM_, W_, NS_, EPI_FP8OUT)
  if (d.stat_out) {  // 66-row panels x 128 columns: to_out without split-K (one balanced round at M = 2112)
    FCFGE(5, 4, 2, 4, EPI_STATS | EPI_FP8OUT)
    return hipErrorInvalidValue;
  }
  if (d.ln_stats) {
    FCFGE(7, 4, 4, 3, EPI_LNFOLD | EPI_FP8OUT) FCFGE(9, 2, 4, 3, EPI_LNFOLD | EPI_FP8OUT)
    FCFGE(17, 2, 4, 2, EPI_LNFOLD | EPI_FP8OUT)
    return hipErrorInvalidValue;
  }
  // 16 waves only where the accumulators leave room under the 128-register cap; tall panels run 8 waves (a 272-row
  // x 256-column tile does not fit 256 registers per lane with 32-byte fragments: callers use <= 208 rows there)
  FCFG(7, 4, 4, 3) FCFG(9, 2, 4, 3) FCFG(9, 4, 2, 3) FCFG(13, 2, 4, 2) FCFG(13, 4, 2, 3) FCFG(17, 4, 2, 3)
  FCFG(17, 2, 4, 2)
#undef FCFG
#undef FCFGE
  return hipErrorInvalidValue;
}

hipError_t igemm_skinny_launch(const GemmDesc& din, int pl, hipStream_t stream) {
  GemmDesc d = din;
  if (d.ksplit < 1) d.ksplit = 1;
  if (PL_COUNT(pl) != 1 || d.taps != 1 || d.in_stride != 1 || d.in_pad != 0 || d.rows_per_b != d.M || d.img_w > 0 ||
      d.M <= 0 || d.M > 128 || d.N <= 0 || d.Cin % 32 != 0 || d.N % 4 != 0 || d.gn_stats || d.stat_out || d.ln_stats ||
      d.out_fp8)
    return hipErrorInvalidValue;
  if (d.swiglu && d.N % 32 != 0) return hipErrorInvalidValue;
  if (d.ksplit > 1 && (!d.out_f32 || d.swiglu || d.ksplit > d.Cin / 32)) return hipErrorInvalidValue;
  const int grid = ((d.N + 31) / 32) * d.ksplit;
  // every M <= 48 runs the 3-sub-tile instantiation (rows beyond M are clamped re-reads, masked in the epilogue): the
  // 1- and 2-sub-tile instantiations of the same source measured 2.5x SLOWER (M = 17: FF-in 35.7 vs 12.4 us), which
  // is what had made the skinny path look like a loss below 33 rows
  int mt = std::max((d.M + 15) / 16, 3);
  if (const char* f = getenv("DSN_SKINNY_MT")) mt = std::max(mt, atoi(f));  // development
#define SK(MT_)                                                                                     \
  if (mt == MT_) {                                                                                  \
    if (PL_F16(pl)) hipLaunchKernelGGL((igemm_skinny_kernel<1, MT_>), dim3(grid), dim3(256), 0, stream, d); \
    else hipLaunchKernelGGL((igemm_skinny_kernel<0, MT_>), dim3(grid), dim3(256), 0, stream, d);     \
    return hipGetLastError();                                                                       \
  }
  SK(3) SK(4) SK(5) SK(6) SK(7) SK(8)
#undef SK
  return hipErrorInvalidValue;
}

hipError_t igemm_panel_launch(const GemmDesc& din, int pl, int bn, hipStream_t stream) {
  const int planes = PL_COUNT(pl), f16 = PL_F16(pl);
  GemmDesc d = din;
  if (d.ksplit < 1) d.ksplit = 1;
  if (d.Cin % 64 != 0 || d.M <= 0 || d.N <= 0 || d.panel_rows <= 0 || d.panel_rows > 17 * 16) return hipErrorInvalidValue;
  if (d.panel_wm == 2 && d.cfg_bk == 32 && d.Cin % 32 != 0) return hipErrorInvalidValue;
  if (d.swiglu && (d.N % 32 != 0)) return hipErrorInvalidValue;
  if (d.ksplit > 1 && (!d.out_f32 || d.swiglu)) return hipErrorInvalidValue;
  if (d.img_w > 0) return hipErrorInvalidValue;
  if (d.stat_out && (d.N % 64 != 0 || d.stat_np != d.N / 64 || d.swiglu || d.ksplit > 1)) return hipErrorInvalidValue;
  if (d.ln_stats && (!d.swiglu || !d.ln_colsum || d.ln_np <= 0 || d.taps != 1 || planes != 1)) return hipErrorInvalidValue;
  const op16_t* zp = zero_page();
  if (!zp) return hipErrorOutOfMemory;
  const int epi = (d.stat_out ? EPI_STATS : 0) | (d.ln_stats ? EPI_LNFOLD : 0);
#define PCFGE(MT_, P_, W_, NS_, BK_, E_)                                                             \
  if (planes == P_ && bn == W_ * 64 && epi == E_)                                                     \
    return f16 ? launch_panel_t<P_, 1, W_, NS_, BK_, MT_, E_>(d, zp, stream)                          \
               : launch_panel_t<P_, 0, W_, NS_, BK_, MT_, E_>(d, zp, stream);
#define PCFGS(MT_, P_, W_, NS_, BK_) PCFGE(MT_, P_, W_, NS_, BK_, 0)
  // 8-wave variants (d.panel_wm == 2): 2 wave rows x 4 wave columns, taller wave tiles; d.cfg_nst / d.cfg_bk pick the ring
#define PCFG2(MT_, NS_, BK_, E_)                                                                                \
  if (planes == 1 && bn == 256 && epi == E_ && d.panel_rows <= MT_ * 16 && (d.cfg_nst == 0 || d.cfg_nst == NS_) && \
      (d.cfg_bk == 0 || d.cfg_bk == BK_))                                                                       \
    return f16 ? launch_panel_t<1, 1, 4, NS_, BK_, MT_, E_, 2>(d, zp, stream)                                   \
               : launch_panel_t<1, 0, 4, NS_, BK_, MT_, E_, 2>(d, zp, stream);
  if (d.panel_wm == 2) {
    PCFG2(9, 3, 64, 0) PCFG2(17, 2, 64, 0) PCFG2(17, 4, 32, 0) PCFG2(17, 2, 64, EPI_LNFOLD) PCFG2(17, 4, 32, EPI_LNFOLD)
  }
#undef PCFG2
  if (d.panel_rows <= 5 * 16) {  // 66-row panels x 128 columns: the N = D residual-stream GEMMs without split-K
    PCFGE(5, 1, 2, 4, 64, EPI_STATS) PCFGS(5, 1, 2, 4, 64)
  }
  if (d.panel_rows <= 7 * 16) {  // 112-row panels (single-plane modes)
    PCFGS(7, 1, 4, 3, 64) PCFGE(7, 1, 4, 3, 64, EPI_LNFOLD)
  }
  if (d.panel_rows <= 9 * 16) {  // short panels (e.g. 16 x 132 rows): 9 row sub-tiles, a 3-stage ring fits
    PCFGS(9, 1, 4, 3, 64) PCFGS(9, 1, 2, 3, 64) PCFGS(9, 2, 4, 2, 32) PCFGS(9, 2, 2, 2, 32)
    PCFGE(9, 1, 4, 3, 64, EPI_LNFOLD)
  }
  if (d.panel_rows <= 13 * 16) {  // 13 sub-tiles (large batches: two rounds of 208-row panels)
    PCFGS(13, 1, 4, 2, 64) PCFGS(13, 1, 2, 2, 64) PCFGE(13, 1, 4, 2, 64, EPI_LNFOLD)
  }
  PCFGS(17, 1, 4, 2, 64) PCFGS(17, 1, 2, 2, 64) PCFGS(17, 2, 4, 2, 32) PCFGS(17, 2, 2, 2, 32)
  PCFGE(17, 1, 4, 2, 64, EPI_LNFOLD)
#undef PCFGS
#undef PCFGE
  return hipErrorInvalidValue;
}
