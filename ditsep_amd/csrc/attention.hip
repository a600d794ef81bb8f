// MFMA attention for the score networks on gfx950: 64-wide DiT heads / one 64-256 wide NCSN++ head.  Sequences up
// to 256 tokens keep every score in registers (attention_mfma_kernel); longer ones walk the keys in blocks with an
// online softmax (attention_long_kernel).
//
// Inputs are the operand planes the fused QKV GEMM epilogue wrote: q (rotary applied,
// pre-scaled by 1/sqrt(dh)) | k (rotary applied) | v, token-major [B*S][3*D].
//
//   scores^T[key][query] = K Q^T     MFMA-A = K rows, MFMA-B = Q rows: both fragments are 16-byte
//                                    row reads straight from global memory, no LDS.
//   softmax over keys                each lane owns ONE query column (lane & 15) and 4 keys per key
//                                    tile (rows 4*(lane>>4) + r): in-lane max/sum + two xor-shuffles.
//   O^T[d][query] = V^T P^T          the un-normalised probabilities stay where the first MFMA left
//                                    them and ARE the B operand of the second one: MFMA k-slot
//                                    (g, jj) is bound to key 16*(2u + jj/4) + 4g + (jj & 3), and the
//                                    V^T fragment is fetched in that same order with
//                                    ds_read_b64_tr_b16 (hardware-transposed LDS read of a
//                                    4-key x 16-feature block) -- no lane movement anywhere.
// Each lane ends with 4 consecutive features of one query -> 8-byte channels-last stores of the
// operand planes the out-projection GEMM consumes.
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// Store one query's output features: this lane holds features dt*16 + 4g + r (dt < NDT) of the query's head.
// 16-bit planes, or -- o8s != null -- fp8 (MX): `out` is then the e4m3 byte tensor [rows][D], o8s the E8M0 scales
// [rows][D/32]; a 32-feature block is a pair of dt tiles across the 4 lane groups that share the query.
template <int P, int F16, int NDT>
__device__ __forceinline__ void store_query_out(op16_t* __restrict__ out, long out_ps, unsigned char* __restrict__ o8s,
                                                long row, int D, int col0, int g, const f32x4 (&oacc)[NDT], float inv) {
  if (o8s) {
    unsigned char* out8 = reinterpret_cast<unsigned char*>(out);
#pragma unroll
    for (int dp = 0; dp < NDT / 2; ++dp) {
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
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *reinterpret_cast<unsigned*>(out8 + row * D + col0 + (2 * dp + u) * 16 + 4 * g) = dsn_fp8x4(v[u] * sc);
      if (g == 0) o8s[row * (D >> 5) + ((col0 + dp * 32) >> 5)] = (unsigned char)(k + 127);
    }
    return;
  }
  const long obase = row * D + col0 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    op16x4 hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      op16_t a, c;
      dsn_split(oacc[dt][r] * inv, a, c, F16);
      hi[r] = a;
      lo[r] = c;
    }
    *reinterpret_cast<op16x4*>(out + obase + dt * 16) = hi;
    if (P == 2) *reinterpret_cast<op16x4*>(out + out_ps + obase + dt * 16) = lo;
  }
}

template <int P, int F16, int NKT, int DH>
__global__ __launch_bounds__(512) void attention_mfma_kernel(const op16_t* __restrict__ qkv, long ps,
                                                             op16_t* __restrict__ out, long out_ps, int S, int H,
                                                             unsigned char* __restrict__ o8s) {
  // blockDim.x / 64 waves per workgroup: they stage V once and each take query tiles of the same (item, head)
  extern __shared__ __attribute__((aligned(16))) op16_t vlds[];  // [P][nkt*16][DH]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * DH;
  const long rs = 3L * D;  // token row stride
  const op16_t* qb = qkv + (long)b * S * rs + h * DH;
  const op16_t* kb = qb + D;
  const op16_t* vb = qb + 2 * D;
  const int nkt = (S + 15) >> 4;
  const int vrows = nkt * 16;

  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;  // role inside a 16-lane transposed-read group
  constexpr int KSD = DH / 32;
  // Short sequences (NKT <= 4) with 64-wide heads are pure latency: issue EVERY global load of the
  // workgroup's single query tile up front (Q, all K tiles, V staging) so they share one round trip.
  constexpr bool PREFETCH = (NKT <= 4 && DH == 64);
  op16x8 pq[P][KSD], pk[PREFETCH ? NKT : 1][P][KSD];
  if (PREFETCH) {
    const int qrow0 = min(((int)blockIdx.y * nwaves + wave) * 16 + r16, S - 1);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks) {
        pq[p][ks] = *reinterpret_cast<const op16x8*>(qb + p * ps + qrow0 * rs + ks * 32 + g * 8);
#pragma unroll
        for (int kt = 0; kt < (PREFETCH ? NKT : 1); ++kt) {
          const int krow = min(kt * 16 + r16, S - 1);
          pk[kt][p][ks] = *reinterpret_cast<const op16x8*>(kb + p * ps + krow * rs + ks * 32 + g * 8);
        }
      }
  }

  // stage V (zero beyond S) -- DH/8 lanes x 16 B per token row
  constexpr int CPRV = DH / 8;
  const op16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int idx = threadIdx.x; idx < vrows * CPRV; idx += blockDim.x) {
    const int row = idx / CPRV, c = idx % CPRV;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const op16x8 v = row < S ? *reinterpret_cast<const op16x8*>(vb + p * ps + row * rs + c * 8) : zero8;
      *reinterpret_cast<op16x8*>(vlds + ((long)p * vrows + row) * DH + c * 8) = v;
    }
  }
  __syncthreads();

  // one query tile per workgroup (blockIdx.y): short sequences are latency bound, so spread the query
  // tiles over more waves (V is staged redundantly, it is small)
  const int qt0 = blockIdx.y * nwaves + wave;
  for (int qt = qt0; qt < nkt; qt += gridDim.y * nwaves) {
    // ---- scores^T = K Q^T -------------------------------------------------
    const int qrow = min(qt * 16 + r16, S - 1);
    op16x8 fq[P][KSD];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks)
        fq[p][ks] = (PREFETCH && qt == qt0)
                        ? pq[p][ks]
                        : *reinterpret_cast<const op16x8*>(qb + p * ps + qrow * rs + ks * 32 + g * 8);
    f32x4 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        const int krow = min(kt * 16 + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          op16x8 fk[P];
#pragma unroll
          for (int p = 0; p < P; ++p)
            fk[p] = PREFETCH ? pk[PREFETCH ? kt : 0][p][ks]
                             : *reinterpret_cast<const op16x8*>(kb + p * ps + krow * rs + ks * 32 + g * 8);
          if (P == 2) {
            sc[kt] = mfma16<F16>(fk[P - 1], fq[0][ks], sc[kt]);
            sc[kt] = mfma16<F16>(fk[0], fq[P - 1][ks], sc[kt]);
          }
          sc[kt] = mfma16<F16>(fk[0], fq[0][ks], sc[kt]);
        }
      }
    }
    // ---- softmax over keys (this lane: query qt*16 + r16, keys kt*16 + 4g + r) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt * 16 + 4 * g + r >= S) sc[kt][r] = -INFINITY;
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
        const float pv = kt < nkt ? expf(sc[kt][r] - mx) : 0.f;  // exp(-inf) = 0 for masked keys
        sc[kt][r] = pv;
        lsum += pv;
      }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);

    // ---- O^T = V^T P^T ----------------------------------------------------------
    constexpr int NDT = DH / 16;
    f32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      if (2 * u < nkt) {
        op16x8 fp[P];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float pv = jj < 4 ? sc[2 * u][jj] : sc[2 * u + 1][jj - 4];
          op16_t hi, lo;
          dsn_split(pv, hi, lo, F16);
          fp[0][jj] = hi;
          if (P == 2) fp[P - 1][jj] = lo;
        }
        const bool second = 2 * u + 1 < nkt;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          op16x8 fv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const op16_t* base = vlds + (long)p * vrows * DH + dt * 16 + 4 * tp;
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base + ((2 * u) * 16 + 4 * g + tq) * DH));
            // rows of a tile past the sequence end are zero-filled in LDS; clamp the address only
            const int t1 = second ? 2 * u + 1 : 2 * u;
            s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base + (t1 * 16 + 4 * g + tq) * DH));
            if (!second) hi4 = s16x4{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fv[p][e] = (unsigned short)lo4[e];
              fv[p][4 + e] = (unsigned short)hi4[e];
            }
          }
          if (P == 2) {
            oacc[dt] = mfma16<F16>(fv[P - 1], fp[0], oacc[dt]);
            oacc[dt] = mfma16<F16>(fv[0], fp[P - 1], oacc[dt]);
          }
          oacc[dt] = mfma16<F16>(fv[0], fp[0], oacc[dt]);
        }
      }
    }
    // ---- normalise + store: query qt*16 + r16, features dt*16 + 4g + r --------
    const int qi = qt * 16 + r16;
    if (qi < S) store_query_out<P, F16, NDT>(out, out_ps, o8s, (long)b * S + qi, D, h * DH, g, oacc, 1.f / lsum);
  }
}

// Long sequences (more than 256 keys: DiT beyond 255 latent frames, NCSN++ attention beyond 64 frames): the same
// fragment scheme, but keys are walked in blocks of KBT tiles with an online softmax (running max m, running sum l,
// accumulator rescaled by exp(m_old - m_new) per block), V staged per block -- registers and LDS no longer grow
// with S.  One wave per (item, head, 16-query tile).
template <int P, int F16, int DH>
__global__ __launch_bounds__(512) void attention_long_kernel(const op16_t* __restrict__ qkv, long ps,
                                                            op16_t* __restrict__ out, long out_ps, int S, int H,
                                                            unsigned char* __restrict__ o8s) {
  constexpr int KBT = 8;         // key tiles per block
  constexpr int KB = KBT * 16;   // keys per block
  extern __shared__ __attribute__((aligned(16))) op16_t vlds[];  // [P][KB][DH]
  // blockDim.x / 64 waves per workgroup: each owns one query tile, all walk the key blocks together and stage each
  // V block once (one wave per workgroup staged the 64 KB block of a 256-wide head alone: 260 us per launch at 944
  // tokens)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * DH;
  const long rs = 3L * D;
  const op16_t* qb = qkv + (long)b * S * rs + h * DH;
  const op16_t* kb = qb + D;
  const op16_t* vb = qb + 2 * D;
  const int nqt = (S + 15) >> 4;
  const int r16 = lane & 15, g = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;
  constexpr int KSD = DH / 32, NDT = DH / 16, CPRV = DH / 8;
  const op16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  for (int qt0 = blockIdx.y * nwaves; qt0 < nqt; qt0 += gridDim.y * nwaves) {
    const int qt = qt0 + wave;  // waves past the last query tile run along (barriers, staging) and store nothing
    const int qrow = min(qt * 16 + r16, S - 1);
    op16x8 fq[P][KSD];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int ks = 0; ks < KSD; ++ks)
        fq[p][ks] = *reinterpret_cast<const op16x8*>(qb + p * ps + qrow * rs + ks * 32 + g * 8);
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < S; k0 += KB) {
      __syncthreads();  // the previous block's V reads are done
      for (int idx = threadIdx.x; idx < KB * CPRV; idx += blockDim.x) {
        const int row = idx / CPRV, c = idx % CPRV;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const op16x8 v = k0 + row < S ? *reinterpret_cast<const op16x8*>(vb + p * ps + (long)(k0 + row) * rs + c * 8) : zero8;
          *reinterpret_cast<op16x8*>(vlds + ((long)p * KB + row) * DH + c * 8) = v;
        }
      }
      __syncthreads();
      // scores^T of this block
      f32x4 sc[KBT];
      float bm = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KBT; ++kt) {
        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int krow = min(k0 + kt * 16 + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          op16x8 fk[P];
#pragma unroll
          for (int p = 0; p < P; ++p)
            fk[p] = *reinterpret_cast<const op16x8*>(kb + p * ps + (long)krow * rs + ks * 32 + g * 8);
          if (P == 2) {
            sc[kt] = mfma16<F16>(fk[P - 1], fq[0][ks], sc[kt]);
            sc[kt] = mfma16<F16>(fk[0], fq[P - 1][ks], sc[kt]);
          }
          sc[kt] = mfma16<F16>(fk[0], fq[0][ks], sc[kt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (k0 + kt * 16 + 4 * g + r >= S) sc[kt][r] = -INFINITY;
          bm = fmaxf(bm, sc[kt][r]);
        }
      }
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float mnew = fmaxf(m, bm);        // every block holds at least one valid key: finite
      const float alpha = expf(m - mnew);     // first block: exp(-inf) = 0
      float bs = 0.f;
#pragma unroll
      for (int kt = 0; kt < KBT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = expf(sc[kt][r] - mnew);
          sc[kt][r] = pv;
          bs += pv;
        }
      bs += __shfl_xor(bs, 16, 64);
      bs += __shfl_xor(bs, 32, 64);
      l = l * alpha + bs;
      m = mnew;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) oacc[dt] *= alpha;
      // O^T += V^T P^T over the block's key-tile pairs
#pragma unroll
      for (int u = 0; u < KBT / 2; ++u) {
        op16x8 fp[P];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float pv = jj < 4 ? sc[2 * u][jj] : sc[2 * u + 1][jj - 4];
          op16_t hi, lo;
          dsn_split(pv, hi, lo, F16);
          fp[0][jj] = hi;
          if (P == 2) fp[P - 1][jj] = lo;
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          op16x8 fv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const op16_t* base = vlds + (long)p * KB * DH + dt * 16 + 4 * tp;
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base + ((2 * u) * 16 + 4 * g + tq) * DH));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base + ((2 * u + 1) * 16 + 4 * g + tq) * DH));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fv[p][e] = (unsigned short)lo4[e];
              fv[p][4 + e] = (unsigned short)hi4[e];
            }
          }
          if (P == 2) {
            oacc[dt] = mfma16<F16>(fv[P - 1], fp[0], oacc[dt]);
            oacc[dt] = mfma16<F16>(fv[0], fp[P - 1], oacc[dt]);
          }
          oacc[dt] = mfma16<F16>(fv[0], fp[0], oacc[dt]);
        }
      }
    }
    const int qi = qt * 16 + r16;
    if (qi < S) store_query_out<P, F16, NDT>(out, out_ps, o8s, (long)b * S + qi, D, h * DH, g, oacc, 1.f / l);
  }
}

template <int P, int F16, int DH>
void launch_t(const op16_t* qkv, long ps, op16_t* out, long out_ps, int B, int S, int H, unsigned char* o8s,
              hipStream_t st) {
  const int nkt = (S + 15) / 16;
  if (nkt > 16) {  // more than 256 keys: blocked keys + online softmax
    const size_t sml = (size_t)P * 128 * DH * sizeof(op16_t);
    static std::atomic<unsigned long long> attr_l{0};
    if (dsn_first_use_on_device(attr_l)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<P, F16, DH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    static const char* wl = getenv("DSN_ATTN_WL");  // development
    const int W = wl ? std::max(1, std::min(8, atoi(wl))) : 4;  // measured (1 / 2 / 4 / 8): NCSN++ 944 tokens 7.48 / 7.17 / 7.14 / 7.31 ms per call, DiT 301 tokens 32.5 / 26.5 / 23.4 / 25.0 us
    hipLaunchKernelGGL((attention_long_kernel<P, F16, DH>), dim3(B * H, (nkt + W - 1) / W), dim3(64 * W), sml, st, qkv, ps,
                       out, out_ps, S, H, o8s);
    return;
  }
  const size_t sm = (size_t)P * nkt * 16 * DH * sizeof(op16_t);
  if (nkt <= 4) {
    // waves per workgroup: each takes one query tile of the same (item, head) and they stage V once
    static const char* wenv = getenv("DSN_ATTN_W");
    const int W = wenv ? std::max(1, std::min(4, atoi(wenv))) : 3;  // measured at S = 33 (3 query tiles): 12.9 / 12.6 / 12.3 / 12.5 us for 1..4
    hipLaunchKernelGGL((attention_mfma_kernel<P, F16, 4, DH>), dim3(B * H, (nkt + W - 1) / W), dim3(64 * W), sm, st, qkv,
                       ps, out, out_ps, S, H, o8s);
  } else {
    static std::atomic<unsigned long long> attr{0};
    if (dsn_first_use_on_device(attr)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel<P, F16, 16, DH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // wide heads stage a large V block (DH * nkt * 32 B): share it between the query tiles' waves
    // the query tiles of an (item, head) share workgroups of up to 8 waves: V is staged once per workgroup (it used to be
    // one wave per workgroup for 64-wide heads: at S = 236 every one of the 15 query tiles staged the 30 KB of V
    // again -- 44 -> 23 us per launch at the C5 shape)
    static const char* wenv16 = getenv("DSN_ATTN_W16");  // development
    int W = nkt >= 8 ? 8 : (nkt >= 4 ? 4 : 1);
    if (wenv16) W = std::max(1, std::min(8, atoi(wenv16)));
    hipLaunchKernelGGL((attention_mfma_kernel<P, F16, 16, DH>), dim3(B * H, (nkt + W - 1) / W), dim3(64 * W), sm, st,
                       qkv, ps, out, out_ps, S, H, o8s);
  }
}

template <int DH>
void launch_dh(const op16_t* qkv, long ps, op16_t* out, long out_ps, int pl, int B, int S, int H, unsigned char* o8s,
               hipStream_t st) {
  const int P = PL_COUNT(pl), f16 = PL_F16(pl);
  if (P == 1 && !f16) launch_t<1, 0, DH>(qkv, ps, out, out_ps, B, S, H, o8s, st);
  else if (P == 2 && !f16) launch_t<2, 0, DH>(qkv, ps, out, out_ps, B, S, H, o8s, st);
  else if (P == 1) launch_t<1, 1, DH>(qkv, ps, out, out_ps, B, S, H, o8s, st);
  else launch_t<2, 1, DH>(qkv, ps, out, out_ps, B, S, H, o8s, st);
}

}  // namespace

// dh = head width (64: DiT; 64/128/256: the single-head NCSN++ attention blocks)
int launch_attention_mfma(const op16_t* qkv, long ps, op16_t* out, long out_ps, int pl, int B, int S, int H, int dh,
                          hipStream_t st, unsigned char* out_fp8_scale) {
  if (dh == 64) launch_dh<64>(qkv, ps, out, out_ps, pl, B, S, H, out_fp8_scale, st);
  else if (dh == 128) launch_dh<128>(qkv, ps, out, out_ps, pl, B, S, H, out_fp8_scale, st);
  else if (dh == 256) launch_dh<256>(qkv, ps, out, out_ps, pl, B, S, H, out_fp8_scale, st);
  else return -1;
  return 0;
}
