// DiT self-attention block front half in ONE launch: to_qkv GEMM + rotary embedding + softmax(Q K^T) V.
// reference: Attention.forward / apply_rotary_pos_emb (src/stable_audio_tools/models/transformer.py:290-598, 92-173).
//
// The DiT of this path attends over T + 1 = 33 tokens per mixture (C2): a separate attention launch is pure latency
// (12.4 us for 0.6 GFLOP) plus a 26 MB round trip of the q | k | v planes per layer.  Here a workgroup owns a row PANEL
// of whole mixtures (ipp items x S tokens <= 144 rows) and ONE head: it computes that head's q | k | v columns
// (192 of the 3 D) for its rows on the MFMA pipe exactly like the row-panel GEMM (global_load_lds ring, counted
// vmcnt, one raw barrier per k-tile), applies the rotary embedding and the 1/sqrt(dh) scale to the accumulators, parks
// q, k, v as 16-bit operands in the LDS the ring no longer needs, and runs the attention of its items from there
// (scores^T = K Q^T, in-lane softmax, O^T = V^T P^T with hardware-transposed V reads: attention.hip's scheme).
// 64 mixtures x 16 heads at S = 33: 16 panels x 16 heads = 256 workgroups = one round of the chip.
//
// Tile: 144 rows x 192 columns, 8 waves as 2 (rows: 5 + 4 sub-tiles) x 4 (48 columns each), BK = 64, NST-stage ring.
// Single-plane 16-bit operand modes only (fp16 / bf16); 64-wide heads.
#include "igemm.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int QA_TN = 192, QA_BK = 64;
constexpr int QA_TM_MAX = 240;                  // tallest panel instantiated (long-form: one 236-token item)

// chunk swizzle of a 128-byte LDS row (see igemm.hip::swzk<64>): conflict-free ds_read_b128 of MFMA fragments
__device__ __forceinline__ int qa_swz(int row) { return (row >> 1) & 7; }
// element offset of (row, column c) inside one [rows][64] operand image
__device__ __forceinline__ int qa_off(int row, int c) { return row * 64 + ((((c >> 3) ^ qa_swz(row))) << 3) + (c & 7); }

// QA_TM: panel rows of the tile -- 144 (9 row sub-tiles as 5 + 4, 3-stage ring) for the 4 s shapes, 240 (15 as 8 + 7,
// 2-stage ring: 111 KB) for one long-form item of up to 240 tokens (BASELINE config 5: 236)
template <int F16, int NST, int NKT, int QA_TM = 144>
__global__ __launch_bounds__(512, 1) void qkv_attention_kernel(const QkvAttnDesc d, const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];  // [NST][QA_TM + 192][64] ring | dummy [8][64]
  constexpr int QA_ROWS = QA_TM + QA_TN;          // staged rows per k-tile: A rows then W rows
  constexpr int QA_STAGE = QA_ROWS * QA_BK;       // elements per ring stage
  constexpr int QA_AG = QA_TM / 8, QA_WG = QA_TN / 8;  // live row groups (8 rows x 128 B each)
  constexpr int QA_GPW = (QA_AG + QA_WG + 7) / 8;      // glds wave-instructions per wave per k-tile (dummies fill up)
  constexpr int MT = QA_TM / 16;                       // row sub-tiles: wave row 0 takes MT0, wave row 1 the rest
  constexpr int MT0 = (MT + 1) / 2, MTW = MT0;
  static_assert(3 * QA_TM * 64 <= NST * QA_STAGE, "the q | k | v images must fit in the ring");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int mtw = wm == 0 ? MT0 : MT - MT0;   // row sub-tiles of this wave
  const int row0 = wm == 0 ? 0 : MT0 * 16;    // first row of this wave inside the panel

  // XCD-aware bijective remap; tiles ordered (head group of 4, panel, head in group): an XCD's contiguous share is a few
  // heads x a few panels, so both its weight columns and its activation panels stay in its L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + loc;
  int head, panel;
  if ((d.H & 3) == 0) {
    const int per_group = d.panels * 4;
    const int hg = t / per_group, rem = t - hg * per_group;
    panel = rem >> 2;
    head = hg * 4 + (rem & 3);
  } else {
    panel = t / d.H;
    head = t - panel * d.H;
  }
  const int S = d.S, K = d.D;
  const int prow = d.ipp * S;                       // rows of a full panel (whole items)
  const int m0 = panel * prow;
  const int rows_here = min(prow, d.M - m0);        // the last panel may hold fewer items
  const int nkt = K / QA_BK;

  // ---- loader: 6 row groups per wave ----
  const int rsub = lane >> 3, cpos = lane & 7;
  const op16_t* src[QA_GPW];
  int step[QA_GPW], dsto[QA_GPW];
#pragma unroll
  for (int gi = 0; gi < QA_GPW; ++gi) {
    const int g = wave * QA_GPW + gi;
    const op16_t* z = zero_page + cpos * 8;
    if (g < QA_AG) {
      const int row = g * 8 + rsub;
      const bool ok = row < rows_here;
      src[gi] = ok ? d.A + (long)(m0 + row) * K + ((cpos ^ qa_swz(row)) << 3) : z;
      step[gi] = ok ? QA_BK : 0;
      dsto[gi] = g * 8 * QA_BK;
    } else if (g < QA_AG + QA_WG) {
      const int r = (g - QA_AG) * 8 + rsub;         // 0..191: q | k | v rows of this head
      const long grow = (long)(r >> 6) * d.D + head * 64 + (r & 63);
      src[gi] = d.W + grow * K + ((cpos ^ qa_swz(r)) << 3);
      step[gi] = QA_BK;
      dsto[gi] = g * 8 * QA_BK;
    } else {
      src[gi] = z;
      step[gi] = 0;
      dsto[gi] = -1;
    }
  }
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int gi = 0; gi < QA_GPW; ++gi) {
      const op16_t* g = src[gi] + (long)kt * step[gi];
      op16_t* dst = dsto[gi] < 0 ? lds + NST * QA_STAGE : lds + stage * QA_STAGE + dsto[gi];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x4 acc[3][MTW];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, g4 = lane >> 4;
  const int fsw = qa_swz(r16);
  const int a_row_off = (row0 + r16) * QA_BK;
  const int w_row_off = (QA_TM + wn * 48 + r16) * QA_BK;

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) issue(s, s);

  for (int i = 0; i < nkt; ++i) {
    // ring protocol of igemm2_kernel: tile i landed once only the younger tiles' loads are outstanding; lgkmcnt(0):
    // this wave's fragment reads of tile i-1 completed before the barrier lets another wave's DMA reuse that stage
    const int younger = min(NST - 2, nkt - 1 - i);
    if (NST >= 4 && younger >= 2)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * QA_GPW) : "memory");
    else if (NST >= 3 && younger >= 1)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QA_GPW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (i + NST - 1 < nkt) issue((i + NST - 1) % NST, i + NST - 1);

    const op16_t* base = lds + (i % NST) * QA_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + g4) ^ fsw) << 3;
      op16x8 fa[MTW], fw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) fw[k] = *reinterpret_cast<const op16x8*>(base + w_row_off + k * 16 * QA_BK + coff);
#pragma unroll
      for (int k = 0; k < MTW; ++k)
        if (k < mtw) fa[k] = *reinterpret_cast<const op16x8*>(base + a_row_off + k * 16 * QA_BK + coff);
#pragma unroll
      for (int tm = 0; tm < MTW; ++tm)
        if (tm < mtw) {
#pragma unroll
          for (int tn = 0; tn < 3; ++tn) acc[tn][tm] = mfma16<F16>(fw[tn], fa[tm], acc[tn][tm]);
        }
    }
  }

  // ---- epilogue 1: bias, rotary embedding, q scale -> 16-bit q | k | v images in LDS (the ring is free now) ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  op16_t* const qkv_lds = lds;  // [3][144][64]
  // rotary pair = (first, first + 1) column sub-tiles of this wave holding features 0..15 / 16..31 of q (wave 0) or k
  // (wave 1: its sub-tile 0 is q 48..63); -1: none
  const int rope_first = wn == 0 ? 0 : (wn == 1 ? 1 : -1);
  const int nq = g4 * 4;
#pragma unroll
  for (int tm = 0; tm < MTW; ++tm) {
    if (tm >= mtw) continue;
    const int row = row0 + tm * 16 + r16;          // row inside the panel
    if (d.bias) {
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) {
        const int col = wn * 48 + tn * 16 + nq;
        acc[tn][tm] += *reinterpret_cast<const f32x4*>(d.bias + (long)(col >> 6) * d.D + head * 64 + (col & 63));
      }
    }
    if (rope_first >= 0) {
      const int pos = row % S;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + nq);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + nq);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(d.rope_cos + pos * 32 + 16 + nq);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(d.rope_sin + pos * 32 + 16 + nq);
      // (wave-uniform choice of the pair, written out so that the accumulator indices stay compile-time constants)
      if (rope_first == 0) {
        const f32x4 x0 = acc[0][tm], x1 = acc[1][tm];
        acc[0][tm] = x0 * c0 - x1 * s0;
        acc[1][tm] = x1 * c1 + x0 * s1;
      } else {
        const f32x4 x0 = acc[1][tm], x1 = acc[2][tm];
        acc[1][tm] = x0 * c0 - x1 * s0;
        acc[2][tm] = x1 * c1 + x0 * s1;
      }
    }
#pragma unroll
    for (int tn = 0; tn < 3; ++tn) {
      const int col = wn * 48 + tn * 16 + nq;       // 0..191
      const int sec = col >> 6, c = col & 63;
      f32x4 v = acc[tn][tm];
      if (sec == 0) v *= d.q_scale;
      op16x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = to_op16(v[r], F16);
      *reinterpret_cast<op16x4*>(qkv_lds + sec * (QA_TM * 64) + qa_off(row, c)) = h;
    }
  }
  __syncthreads();

  // ---- epilogue 2: attention of the panel's items, one (item, query tile) job per wave at a time ----
  const op16_t* const ql = qkv_lds;
  const op16_t* const kl = qkv_lds + QA_TM * 64;
  const op16_t* const vl = qkv_lds + 2 * QA_TM * 64;
  const int nqt = (S + 15) >> 4;
  const int items = rows_here / S;
  const int tq = r16 >> 2, tp = r16 & 3;            // role inside a 16-lane transposed-read group
  for (int job = wave; job < items * nqt; job += 8) {
    const int it = job / nqt, qt = job - it * nqt;
    const int base = it * S;
    const int qrow = base + min(qt * 16 + r16, S - 1);
    op16x8 fq[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fq[ks] = *reinterpret_cast<const op16x8*>(ql + qa_off(qrow, (ks * 4 + g4) * 8));
    f32x4 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nqt) {
        const int krow = base + min(kt * 16 + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const op16x8 fk = *reinterpret_cast<const op16x8*>(kl + qa_off(krow, (ks * 4 + g4) * 8));
          sc[kt] = mfma16<F16>(fk, fq[ks], sc[kt]);
        }
      }
    }
    // softmax over keys: this lane owns query qt*16 + r16 and keys kt*16 + 4 g4 + r
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nqt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt * 16 + 4 * g4 + r >= S) sc[kt][r] = -INFINITY;
          mx = fmaxf(mx, sc[kt][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = kt < nqt ? expf(sc[kt][r] - mx) : 0.f;
        sc[kt][r] = pv;
        lsum += pv;
      }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);

    // O^T = V^T P^T: MFMA k-slot (g4, jj) <-> key 16 (2u + jj/4) + 4 g4 + (jj & 3); V^T by transposed LDS reads.  Keys
    // past S carry probability 0, so their V rows (the next item's, or clamped to the image) only have to be readable.
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (NKT + 1) / 2; ++u) {
      if (2 * u < nqt) {
        op16x8 fp;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float pv = jj < 4 ? sc[2 * u][jj] : (2 * u + 1 < NKT ? sc[(2 * u + 1 < NKT) ? 2 * u + 1 : 0][jj - 4] : 0.f);
          fp[jj] = to_op16(pv, F16);
        }
        const int v0 = min(base + (2 * u) * 16 + 4 * g4 + tq, QA_TM - 1);
        const int v1 = min(base + (2 * u + 1) * 16 + 4 * g4 + tq, QA_TM - 1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vl + qa_off(v0, dt * 16 + 4 * tp)));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vl + qa_off(v1, dt * 16 + 4 * tp)));
          op16x8 fv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            fv[e] = (unsigned short)lo4[e];
            fv[4 + e] = (unsigned short)hi4[e];
          }
          oacc[dt] = mfma16<F16>(fv, fp, oacc[dt]);
        }
      }
    }
    // query qt*16 + r16, features dt*16 + 4 g4 + r of this head -> the out-projection's operand plane
    const float inv = 1.f / lsum;
    const long orow = (long)(m0 + base + qt * 16 + r16);
    if (d.out8) {
      // fp8 (MX) output for the fp8 out-projection: a 32-feature scale block = a pair of feature tiles across the 4 lane
      // groups that share the query (attention.hip::store_query_out); the shuffles run for every lane, stores are masked
#pragma unroll
      for (int dp = 0; dp < 2; ++dp) {
        f32x4 v[2];
        float amax = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          v[u] = oacc[2 * dp + u] * inv;
#pragma unroll
          for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(v[u][r]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int k = dsn_mx_exp(amax);
        const float sc = dsn_pow2(-k);
        if (qt * 16 + r16 < S) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
            *reinterpret_cast<unsigned*>(d.out8 + orow * d.D + head * 64 + (2 * dp + u) * 16 + 4 * g4) = dsn_fp8x4(v[u] * sc);
          if (g4 == 0) d.out8_scale[orow * (d.D >> 5) + ((head * 64 + dp * 32) >> 5)] = (unsigned char)(k + 127);
        }
      }
    } else if (qt * 16 + r16 < S) {
      op16_t* o = d.out + orow * d.D + head * 64 + 4 * g4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        op16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = to_op16(oacc[dt][r] * inv, F16);
        *reinterpret_cast<op16x4*>(o + dt * 16) = h;
      }
    }
  }
}

const op16_t* qa_zero_page() {
  static op16_t* zp[64] = {};
  op16_t*& z = zp[dsn_current_device()];
  if (!z) {
    if (hipMalloc((void**)&z, 4096) != hipSuccess) return nullptr;
    (void)hipMemset(z, 0, 4096);
  }
  return z;
}

template <int F16, int NST, int NKT, int TM>
hipError_t qa_launch_t(const QkvAttnDesc& d, const op16_t* zp, hipStream_t stream) {
  static std::atomic<unsigned long long> attr{0};
  if (dsn_first_use_on_device(attr))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_attention_kernel<F16, NST, NKT, TM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t smem = (size_t)(NST * (TM + QA_TN) * QA_BK + 8 * QA_BK) * sizeof(op16_t);
  hipLaunchKernelGGL((qkv_attention_kernel<F16, NST, NKT, TM>), dim3(d.panels * d.H), dim3(512), smem, stream, d, zp);
  return hipGetLastError();
}

}  // namespace

int qkv_attention_max_rows() { return QA_TM_MAX; }
int qkv_attention_panel_rows(int rows) { return rows <= 144 ? 144 : QA_TM_MAX; }

hipError_t qkv_attention_launch(const QkvAttnDesc& din, int pl, hipStream_t stream) {
  QkvAttnDesc d = din;
  if (PL_COUNT(pl) != 1 || d.D != d.H * 64 || d.D % QA_BK != 0 || d.S < 1 || d.ipp < 1 || d.ipp * d.S > QA_TM_MAX ||
      d.M <= 0 || d.M % d.S != 0 || !d.A || !d.W || (!d.out && !d.out8) || (d.out8 && !d.out8_scale) || !d.rope_cos ||
      !d.rope_sin || d.A == d.out)
    return hipErrorInvalidValue;
  d.panels = cdiv(d.M / d.S, d.ipp);
  const op16_t* zp = qa_zero_page();
  if (!zp) return hipErrorOutOfMemory;
  const int nqt = (d.S + 15) / 16;
  const int f16 = PL_F16(pl);
#define QA(NST_, NKT_, TM_) \
  return f16 ? qa_launch_t<1, NST_, NKT_, TM_>(d, zp, stream) : qa_launch_t<0, NST_, NKT_, TM_>(d, zp, stream);
  if (d.ipp * d.S <= 144) {
    if (nqt <= 3) { QA(3, 3, 144) }
    if (nqt <= 5) { QA(3, 5, 144) }
    QA(3, 9, 144)
  }
  // tall panels (up to 240 rows, e.g. one 30 s item of 236 tokens): 2-stage ring
  if (nqt <= 9) { QA(2, 9, 240) }
  QA(2, 15, 240)
#undef QA
}
