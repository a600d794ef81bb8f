// Non-GEMM kernels of the separation path for gfx950: layout transforms, the
// predictor-corrector elementwise updates, residual + LayerNorm, Oobleck edge
// convolutions, SI-SDR sums, Philox RNG and weight packing (attention: attention.hip).
// All are HBM/LDS-bound byte movers: coalesced 16-byte accesses, wave64 shuffles
// for reductions, no MFMA.
#include "kernels.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(long n, int per_block = TPB, int cap = 256 * 16) {
  long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// `planes` everywhere in this file is the packed DSN_PL(plane count, fp16 flag)
__device__ __forceinline__ void store_planes(op16_t* dst, long ps, int planes, long i, float v) {
  op16_t h, l;
  dsn_split(v, h, l, PL_F16(planes));
  dst[i] = h;
  if (PL_COUNT(planes) == 2) dst[ps + i] = l;
}

// ------------------------------------------------------------------ transforms
__global__ void pack_tokens_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1,
                                   int B, int T, float* __restrict__ df, op16_t* __restrict__ dp, long ps,
                                   int planes) {
  const int C = C0 + C1;
  const long n = (long)B * T * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const float v = c < C0 ? s0[((long)b * C0 + c) * T + t] : s1[((long)b * C1 + (c - C0)) * T + t];
    if (df) df[i] = v;
    if (dp) store_planes(dp, ps, planes, i, v);
  }
}

// dst[r * dst_stride + c] = src[r * width + c]  (width % 4 == 0): the timestep-token rows into the residual stream
__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int width4,
                                 long dst_stride) {
  const long n = (long)rows * width4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / width4;
    const int c = (int)(i - r * width4);
    reinterpret_cast<f32x4*>(dst + r * dst_stride)[c] = reinterpret_cast<const f32x4*>(src + r * (long)width4 * 4)[c];
  }
}

__global__ void unpack_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int T) {
  const long n = (long)B * C * T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long bc = i / T;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    dst[i] = src[((long)b * T + t) * C + c];
  }
}

__global__ void to_planes_kernel(const float* __restrict__ src, op16_t* __restrict__ dst, long ps, int planes,
                                 long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    op16x4 hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      op16_t h, l;
      dsn_split(v[r], h, l, PL_F16(planes));
      hi[r] = h;
      lo[r] = l;
    }
    reinterpret_cast<op16x4*>(dst)[i] = hi;
    if (PL_COUNT(planes) == 2) reinterpret_cast<op16x4*>(dst + ps)[i] = lo;
  }
}

// ------------------------------------------------------------------ PC sampler
// index helpers for x[B,n,D,T] (i linear), y[B,1,D,T], score token-major [B*T][n*D]
struct PcIdx {
  long yi, si;
};
__device__ __forceinline__ PcIdx pc_index(long i, int n, int D, int T) {
  const int t = (int)(i % T);
  long r = i / T;
  const int c = (int)(r % D);
  r /= D;
  const int s = (int)(r % n);
  const long b = r / n;
  PcIdx o;
  o.yi = (b * D + c) * T + t;
  o.si = (b * T + t) * ((long)n * D) + (long)s * D + c;
  return o;
}

// mean_full: the prior mean is a full [B,n,D,T] tensor (`true_mean`, sdes/__init__.py:175-176), else y broadcast
__global__ void pc_prior_kernel(const float* __restrict__ y, const float* __restrict__ z, float* __restrict__ x,
                                float stdT, int n, int D, int T, long total, int mean_full) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const PcIdx ix = pc_index(i, n, D, T);
    x[i] = y[mean_full ? i : ix.yi] + z[i] * stdT;
  }
}

// x_mean = x + step * score ; x = x_mean + gain * z.  ALD passes host scalars (step_dev == null); the Langevin
// corrector's step comes from the per-item norms: step = 2 (snr * mean|z| / mean|score|)^2 (correctors.py:46-53),
// every block re-derives the two batch means in the same fixed order.
__global__ void pc_corrector_kernel(float* __restrict__ x, float* __restrict__ xmean, const float* __restrict__ sc,
                                    const float* __restrict__ z, float step, float gain,
                                    const float* __restrict__ norms, int B, float snr, int n, int D, int T,
                                    long total) {
  if (norms) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
      float gs = 0.f, ns = 0.f;
      for (int b = 0; b < B; ++b) {
        gs += norms[b];
        ns += norms[B + b];
      }
      const float q = snr * (ns / (float)B) / (gs / (float)B);
      sh[0] = q * q * 2.f;
      sh[1] = sqrtf(sh[0] * 2.f);
    }
    __syncthreads();
    step = sh[0];
    gain = sh[1];
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const PcIdx ix = pc_index(i, n, D, T);
    const float xm = x[i] + step * sc[ix.si];
    if (xmean) xmean[i] = xm;
    x[i] = xm + z[i] * gain;
  }
}

// out[b] = ||a[b, :]||_2 over `per_item` contiguous floats; one workgroup per item, fixed reduction order
__global__ void pc_item_norm_kernel(const float* __restrict__ a, long per_item, float* __restrict__ out) {
  const float* p = a + (long)blockIdx.x * per_item;
  float s = 0.f;
  for (long i = threadIdx.x; i < per_item; i += blockDim.x) s += p[i] * p[i];
  s = wave_sum(s);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    out[blockIdx.x] = sqrtf(t);
  }
}

// reverse diffusion (em == 0): x_mean = x - (theta (y-x) dt - G^2 s),            x = x_mean + G z,   G = g sqrt(dt)
// Euler-Maruyama   (em == 1): x_mean = x + (theta (y-x) - g^2 s) (-dt),          x = x_mean + G z
__global__ void pc_predictor_kernel(float* __restrict__ x, float* __restrict__ xmean, const float* __restrict__ y,
                                    const float* __restrict__ sc, const float* __restrict__ z, float theta,
                                    float dt, float G, float g, int em, int n, int D, int T, long total) {
  const float G2 = G * G, g2 = g * g;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const PcIdx ix = pc_index(i, n, D, T);
    const float xv = x[i];
    float xm;
    if (em) {
      const float total_drift = theta * (y[ix.yi] - xv) + (-g2 * sc[ix.si]);
      xm = xv + total_drift * (-dt);
    } else {
      const float f = theta * (y[ix.yi] - xv) * dt;
      const float rev = f - G2 * sc[ix.si];
      xm = xv - rev;
    }
    xmean[i] = xm;
    x[i] = xm + G * z[i];
  }
}

// ------------------------------------------------------------------ secondary sampler family on the latent state
// MixSDE / PriorMixSDE (+ the ald2 corrector) and the Schroedinger-bridge sampler of the reference's sdes package
// (src/sdes/sdes.py:182-593,701-779, correctors.py:87-121, __init__.py:284-389), on x [B,n,D,T] read as [B,n,D*T].
// With A = 11^T/n and Pn = I - A every matrix of these SDEs is a*A + p*Pn:  (aA + pPn) v = a*mean_src(v) +
// p*(v - mean_src(v)).  One thread per (b, d, t) position walks the n <= 4 sources; smix [B][D*T] is PriorMixSDE's
// running RMS of the mixture (null = 1).
__global__ void sigma_mix_kernel(const float* __restrict__ y, float* __restrict__ smix, int L, int avg_len, long total) {
  // 0.5 * sqrt(clamp(avg_pool1d(y^2, avg_len, stride 1, zero padding avg_len/2, count_include_pad), 1e-4))
  const int pad = avg_len / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / L;
    const int l = (int)(i - b * L);
    float acc = 0.f;
    for (int k = 0; k < avg_len; ++k) {
      const int j = l - pad + k;
      if (j >= 0 && j < L) {
        const float v = y[b * L + j];
        acc += v * v;
      }
    }
    smix[i] = 0.5f * sqrtf(fmaxf(acc / (float)avg_len, 1e-4f));
  }
}

__global__ void mix_prior_kernel(const float* __restrict__ y, const float* __restrict__ z, float* __restrict__ x,
                                 const float* __restrict__ smix, float s1, float s2, int n, int D, int T, long npos) {
  const long DT = (long)D * T;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npos; p += (long)gridDim.x * blockDim.x) {
    const long b = p / DT, l = p - b * DT;
    const float sm = smix ? smix[p] : 1.f;
    float mz = 0.f;
    for (int s = 0; s < n; ++s) mz += z[(b * n + s) * DT + l];
    mz /= (float)n;
    const float mean = 0.5f * y[p];
    for (int s = 0; s < n; ++s) {
      const long i = (b * n + s) * DT + l;
      x[i] = mean + (s1 * sm) * mz + (s2 * sm) * (z[i] - mz);
    }
  }
}

// ald2: grad = L L score, x_mean = x + 2 snr^2 grad, x = x_mean + 2 snr L z with L = sqrt(ev1) A + sqrt(ev2) Pn (x smix)
__global__ void mix_corrector_kernel(float* __restrict__ x, float* __restrict__ xmean, const float* __restrict__ sc,
                                     const float* __restrict__ z, const float* __restrict__ smix, float sq1, float sq2,
                                     float snr, int n, int D, int T, long npos) {
  const long DT = (long)D * T;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npos; p += (long)gridDim.x * blockDim.x) {
    const long b = p / DT, l = p - b * DT;
    const int c = (int)(l / T), t = (int)(l - (long)c * T);
    const float sm = smix ? smix[p] : 1.f;
    const float a1 = sq1 * sm, a2 = sq2 * sm;
    float sv[4], zv[4], ms = 0.f, mz = 0.f;
    for (int s = 0; s < n; ++s) {
      sv[s] = sc[(b * T + t) * ((long)n * D) + (long)s * D + c];
      zv[s] = z[(b * n + s) * DT + l];
      ms += sv[s];
      mz += zv[s];
    }
    ms /= (float)n;
    mz /= (float)n;
    // first L: u = a1 ms + a2 (s - ms); its source mean is a1 ms; second L: a1 (a1 ms) + a2 (u - a1 ms)
    for (int s = 0; s < n; ++s) {
      const long i = (b * n + s) * DT + l;
      const float u = a1 * ms + a2 * (sv[s] - ms);
      const float grad = a1 * (a1 * ms) + a2 * (u - a1 * ms);
      const float xm = x[i] + 2.f * snr * snr * grad;
      if (xmean) xmean[i] = xm;
      x[i] = xm + (2.f * snr * a1) * mz + (2.f * snr * a2) * (zv[s] - mz);
    }
  }
}

// reverse diffusion (em = 0): x_mean = x + lambda dt (x - m) + G^2 s, x = x_mean + G z with G = g sqrt(dt) smix;
// Euler-Maruyama (em = 1): x_mean = x + (lambda (x - m) + (g smix)^2 s) dt, x = x_mean + g smix sqrt(dt) z
__global__ void mix_predictor_kernel(float* __restrict__ x, float* __restrict__ xmean, const float* __restrict__ sc,
                                     const float* __restrict__ z, const float* __restrict__ smix, float lambda, float dt,
                                     float g, float sqdt, int em, int n, int D, int T, long npos) {
  const long DT = (long)D * T;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npos; p += (long)gridDim.x * blockDim.x) {
    const long b = p / DT, l = p - b * DT;
    const int c = (int)(l / T), t = (int)(l - (long)c * T);
    const float sm = smix ? smix[p] : 1.f;
    float xv[4], mx = 0.f;
    for (int s = 0; s < n; ++s) {
      xv[s] = x[(b * n + s) * DT + l];
      mx += xv[s];
    }
    mx /= (float)n;
    const float gs = g * sm;
    for (int s = 0; s < n; ++s) {
      const long i = (b * n + s) * DT + l;
      const float scv = sc[(b * T + t) * ((long)n * D) + (long)s * D + c];
      float xm, xn;
      if (em) {
        const float total = -lambda * (xv[s] - mx) - gs * gs * scv;
        xm = xv[s] + total * (-dt);
        xn = xm + gs * sqdt * z[i];
      } else {
        const float f = -lambda * (xv[s] - mx) * dt;
        const float G = gs * sqdt;
        const float rev = f - G * G * scv;
        xm = xv[s] - rev;
        xn = xm + G * z[i];
      }
      xmean[i] = xm;
      x[i] = xn;
    }
  }
}

// Schroedinger-bridge step: x = w_prev x + w_est est + w3 * (third_is_y ? y : z)      (est token-major)
__global__ void sb_update_kernel(float* __restrict__ x, const float* __restrict__ est, const float* __restrict__ third,
                                 float w_prev, float w_est, float w3, int third_is_y, int n, int D, int T, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const PcIdx ix = pc_index(i, n, D, T);
    const float th = third ? third[third_is_y ? ix.yi : i] : 0.f;
    x[i] = w_prev * x[i] + w_est * est[ix.si] + w3 * th;
  }
}
// x[b, s] = y[b, 0] for every source s   (xt = y.repeat(1, n, 1, 1))
__global__ void repeat_sources_kernel(const float* __restrict__ y, float* __restrict__ x, int n, int D, int T, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    x[i] = y[pc_index(i, n, D, T).yi];
}

// ------------------------------------------------------------------ LayerNorm
// Residual-stream update fused with the next LayerNorm: x[row] += bias + sum of the split-K
// partial slabs of the preceding GEMM (written back when any were added), then LayerNorm
// (or a plain copy) to operand planes.  One wave per row, the row lives in registers.
// NS = number of slabs at compile time: every load of a quad (x, bias, all slabs) is issued before the first add.
// BATCH (few rows: single mixtures, where the kernel is one latency chain): x, bias and the slabs of ALL the lane's quads
// are issued before the first add, the sums are written back after the last load, and gamma / beta of all quads are
// requested before the reductions start -- about three memory round trips per row where the quad-by-quad form takes
// twelve (its in-place store of quad k keeps the loads of quad k+1 behind it).  It needs up to 256 VGPRs, which costs
// the bandwidth-bound large-M case occupancy (M = 2112: 18.8 vs 15.9 us), hence the switch.
template <int MAXV, int NS, bool BATCH = false>
__global__ __launch_bounds__(TPB) void residual_norm_kernel(float* __restrict__ x, const float* __restrict__ slabs,
                                                            int nslab, long slab_stride, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, op16_t* __restrict__ out,
                                                            long ps, int planes, int rows, int D, float eps, int do_norm,
                                                            unsigned char* __restrict__ o8s) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int nv = D >> 2;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
    const long rbase = (long)row * D;
    f32x4 v[MAXV];
    f32x4 gq[BATCH ? MAXV : 1], bq[BATCH ? MAXV : 1];
    float s = 0.f;
    if constexpr (BATCH) {
      f32x4 bv[MAXV];
      f32x4 part[NS > 0 ? NS : 1][MAXV];
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int i = min(lane + k * 64, nv - 1);  // clamped: lanes past the row re-read its last quad, never stored
        v[k] = reinterpret_cast<const f32x4*>(x + rbase)[i];
        bv[k] = (NS > 0 && bias) ? reinterpret_cast<const f32x4*>(bias)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int z = 0; z < NS; ++z) part[z][k] = reinterpret_cast<const f32x4*>(slabs + z * slab_stride + rbase)[i];
      }
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int i = lane + k * 64;
        if (do_norm) {
          gq[k] = reinterpret_cast<const f32x4*>(gamma)[min(i, nv - 1)];
          bq[k] = beta ? reinterpret_cast<const f32x4*>(beta)[min(i, nv - 1)] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        v[k] += bv[k];
#pragma unroll
        for (int z = 0; z < NS; ++z) v[k] += part[z][k];
        if (i < nv) {
          if (NS > 0) reinterpret_cast<f32x4*>(x + rbase)[i] = v[k];
          s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int i = lane + k * 64;
        if (i < nv) {
          f32x4 a = reinterpret_cast<const f32x4*>(x + rbase)[i];
          if (NS > 0) {
            f32x4 part[NS > 0 ? NS : 1];
#pragma unroll
            for (int z = 0; z < NS; ++z) part[z] = reinterpret_cast<const f32x4*>(slabs + z * slab_stride + rbase)[i];
            if (bias) a += reinterpret_cast<const f32x4*>(bias)[i];
#pragma unroll
            for (int z = 0; z < NS; ++z) a += part[z];
            reinterpret_cast<f32x4*>(x + rbase)[i] = a;
          }
          v[k] = a;
          s += (a[0] + a[1]) + (a[2] + a[3]);
        }
      }
    }
    float mean = 0.f, rstd = 1.f;
    if (do_norm) {
      mean = wave_sum(s) / D;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int i = lane + k * 64;
        if (i < nv) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dlt = v[k][r] - mean;
            q += dlt * dlt;
          }
        }
      }
      rstd = rsqrtf(wave_sum(q) / D + eps);
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int i = lane + k * 64;
      if (i < nv) {
        f32x4 o = v[k];
        if (do_norm) {
          if constexpr (BATCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * gq[k][r] + bq[k][r];
          } else {
            const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[k][r] - mean) * rstd * g[r];
            if (beta) o += reinterpret_cast<const f32x4*>(beta)[i];
          }
        }
        if (o8s) {
          // fp8 (MX) output: a 32-column scale block = the quads of 8 consecutive lanes (nv % 8 == 0: whole groups)
          float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
          amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
          const int kx = dsn_mx_exp(amax);
          reinterpret_cast<unsigned*>(out)[(rbase >> 2) + i] = dsn_fp8x4(o * dsn_pow2(-kx));
          if ((lane & 7) == 0) o8s[(rbase >> 5) + (i >> 3)] = (unsigned char)(kx + 127);
          continue;
        }
        op16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          op16_t h, l;
          dsn_split(o[r], h, l, PL_F16(planes));
          hi[r] = h;
          lo[r] = l;
        }
        const long oi = (rbase >> 2) + i;
        reinterpret_cast<op16x4*>(out)[oi] = hi;
        if (PL_COUNT(planes) == 2) reinterpret_cast<op16x4*>(out + ps)[oi] = lo;
      }
    }
  }
}

// Large-M form for D = 4 * TPB (the DiT's 1024): ONE ROW PER WORKGROUP, one quad per thread.  With a wave per row the
// C2 shape (2112 rows) put 8 waves on a CU, each with five 16-byte loads in flight -- 40 KB per CU, well short of what
// HBM needs to stay busy; here a CU holds ~33 waves and every load of a row is issued at once.  The row statistics go
// through LDS in wave order (fixed order: bit-reproducible).
template <int NS>
__global__ __launch_bounds__(TPB) void residual_norm_row_kernel(float* __restrict__ x, const float* __restrict__ slabs,
                                                                long slab_stride, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, op16_t* __restrict__ out,
                                                                long ps, int planes, int D, float eps, int do_norm,
                                                                unsigned char* __restrict__ o8s) {
  __shared__ float red[2][TPB / 64];
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
  const long rbase = (long)blockIdx.x * D;
  f32x4 a = reinterpret_cast<const f32x4*>(x + rbase)[i];
  f32x4 part[NS > 0 ? NS : 1];
#pragma unroll
  for (int z = 0; z < NS; ++z) part[z] = reinterpret_cast<const f32x4*>(slabs + z * slab_stride + rbase)[i];
  f32x4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
  if (do_norm) {
    g = reinterpret_cast<const f32x4*>(gamma)[i];
    if (beta) be = reinterpret_cast<const f32x4*>(beta)[i];
  }
  if (NS > 0) {
    if (bias) a += reinterpret_cast<const f32x4*>(bias)[i];
#pragma unroll
    for (int z = 0; z < NS; ++z) a += part[z];
    reinterpret_cast<f32x4*>(x + rbase)[i] = a;
  }
  f32x4 o = a;
  if (do_norm) {
    const float s = wave_sum((a[0] + a[1]) + (a[2] + a[3]));
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) tot += red[0][w];
    const float mean = tot / D;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dlt = a[r] - mean;
      q += dlt * dlt;
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) qt += red[1][w];
    const float rstd = rsqrtf(qt / D + eps);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (a[r] - mean) * rstd * g[r] + be[r];
  }
  if (o8s) {  // fp8 (MX) output: a 32-column scale block = the quads of 8 consecutive lanes
    float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const int kx = dsn_mx_exp(amax);
    reinterpret_cast<unsigned*>(out)[(rbase >> 2) + i] = dsn_fp8x4(o * dsn_pow2(-kx));
    if ((lane & 7) == 0) o8s[(rbase >> 5) + (i >> 3)] = (unsigned char)(kx + 127);
    return;
  }
  op16x4 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    op16_t h, l;
    dsn_split(o[r], h, l, PL_F16(planes));
    hi[r] = h;
    lo[r] = l;
  }
  const long oi = (rbase >> 2) + i;
  reinterpret_cast<op16x4*>(out)[oi] = hi;
  if (PL_COUNT(planes) == 2) reinterpret_cast<op16x4*>(out + ps)[oi] = lo;
}

__global__ void timestep_features_kernel(const float* __restrict__ t, const float* __restrict__ w, int B, int half,
                                         op16_t* __restrict__ out, long ps, int planes) {
  const int n = B * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int b = i / half, k = i - b * half;
    const float f = 2.f * 3.14159265358979323846f * t[b] * w[k];
    store_planes(out, ps, planes, (long)b * 2 * half + k, cosf(f));
    store_planes(out, ps, planes, (long)b * 2 * half + half + k, sinf(f));
  }
}

__global__ void rope_tables_kernel(float* __restrict__ ct, float* __restrict__ st, int S, int rot) {
  const int half = rot / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S * rot; i += gridDim.x * blockDim.x) {
    const int p = i / rot, d = i - p * rot;
    const int k = d % half;
    const float inv = 1.0f / powf(10000.0f, (float)(2 * k) / (float)rot);
    const float f = (float)p * inv;
    ct[i] = cosf(f);
    st[i] = sinf(f);
  }
}

// ------------------------------------------------------------------ Oobleck edges
// Cout == 1 convolution over already-activated planes: 64 outputs per workgroup,
// (64 + ktaps - 1) rows staged in LDS (fp32, rows padded by 4 floats), 4 channel
// quarters reduced through LDS.
__global__ __launch_bounds__(256) void conv_out1_kernel(const op16_t* __restrict__ a, long ps, int planes,
                                                        const float* __restrict__ w, float* __restrict__ out,
                                                        int L, int C, int ktaps, int apply_tanh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RS = C + 4;
  const int nrows = 64 + ktaps - 1;
  float* rows = smem;              // [nrows][RS]
  float* wl = rows + nrows * RS;   // [ktaps][C]
  float* part = wl + ktaps * C;    // [4][64]
  const int blocks_per_seq = (L + 63) / 64;
  const int s = blockIdx.x / blocks_per_seq;
  const int l0 = (blockIdx.x - s * blocks_per_seq) * 64;
  const int pad = (ktaps - 1) / 2;
  const int c8 = C / 8;
  for (int i = threadIdx.x; i < nrows * c8; i += blockDim.x) {
    const int r = i / c8, ch = (i - r * c8) * 8;
    const int l = l0 - pad + r;
    float v[8];
    if (l >= 0 && l < L) {
      const long gi = ((long)s * L + l) * C + ch;
      const op16x8 hi = *reinterpret_cast<const op16x8*>(a + gi);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = from_op16(hi[k], PL_F16(planes));
      if (PL_COUNT(planes) == 2) {
        const op16x8 lo = *reinterpret_cast<const op16x8*>(a + ps + gi);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += from_op16(lo[k], PL_F16(planes));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[r * RS + ch + k] = v[k];
  }
  for (int i = threadIdx.x; i < ktaps * C; i += blockDim.x) wl[i] = w[i];
  __syncthreads();
  const int pos = threadIdx.x & 63, cq = threadIdx.x >> 6;
  const int cw = C / 4;
  float acc = 0.f;
  for (int t = 0; t < ktaps; ++t) {
    const float* rr = rows + (pos + t) * RS + cq * cw;
    const float* ww = wl + t * C + cq * cw;
    for (int c = 0; c < cw; c += 4) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(rr + c);
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(ww + c);
      acc += x4[0] * w4[0] + x4[1] * w4[1] + x4[2] * w4[2] + x4[3] * w4[3];
    }
  }
  part[cq * 64 + pos] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int l = l0 + threadIdx.x;
    if (l < L) {
      float v = (part[threadIdx.x] + part[64 + threadIdx.x]) + (part[128 + threadIdx.x] + part[192 + threadIdx.x]);
      out[(long)s * L + l] = apply_tanh ? tanhf(v) : v;
    }
  }
}

// Same convolution, each input row read ONCE (C % 32 == 0, at most 8 taps): 4 lanes per row take interleaved
// 8-channel chunks (64 contiguous bytes per row per load), form the row's KT per-tap partial dot products against
// weights broadcast from LDS, reduce over the 4 lanes, and a second pass adds the KT shifted partials per output.
// The first version re-read every staged row KT times from LDS (LDS bound, 22 % of the HBM roofline).
template <int KT>
__global__ __launch_bounds__(256) void conv_out1_rows_kernel(const op16_t* __restrict__ a, long ps, int planes,
                                                             const float* __restrict__ w, float* __restrict__ out,
                                                             int L, int C, int apply_tanh) {
  constexpr int TILE = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NJ = C / 32;
  f32x4* wl4 = reinterpret_cast<f32x4*>(smem);  // [KT][NJ][2][4 lanes] float4
  float* part = smem + KT * C;                  // [TILE + KT - 1][8]
  const int blocks_per_seq = (L + TILE - 1) / TILE;
  const int s = blockIdx.x / blocks_per_seq;
  const int l0 = (blockIdx.x - s * blocks_per_seq) * TILE;
  const int pad = (KT - 1) / 2;
  for (int i = threadIdx.x; i < KT * C; i += blockDim.x) {
    const int t = i / C, c = i - t * C;
    const int chunk = c >> 3, k = c & 7;
    const int j = chunk >> 2, q = chunk & 3;
    smem[((((t * NJ + j) * 2 + (k >> 2)) * 4 + q) << 2) + (k & 3)] = w[i];
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
  const bool f16 = PL_F16(planes), two = PL_COUNT(planes) == 2;
  for (int r = threadIdx.x >> 2; r < TILE + KT - 1; r += 64) {
    const int l = l0 - pad + r;
    float acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) acc[t] = 0.f;
    if (l >= 0 && l < L) {
      const op16_t* row = a + ((long)s * L + l) * C;
      for (int j = 0; j < NJ; ++j) {
        const int ch = (j * 4 + q) * 8;
        const op16x8 hi = *reinterpret_cast<const op16x8*>(row + ch);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = from_op16(hi[k], f16);
        if (two) {
          const op16x8 lo = *reinterpret_cast<const op16x8*>(row + ps + ch);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += from_op16(lo[k], f16);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const f32x4 wa = wl4[((t * NJ + j) * 2 + 0) * 4 + q], wb = wl4[((t * NJ + j) * 2 + 1) * 4 + q];
          acc[t] += (v[0] * wa[0] + v[1] * wa[1]) + (v[2] * wa[2] + v[3] * wa[3]) + (v[4] * wb[0] + v[5] * wb[1]) +
                    (v[6] * wb[2] + v[7] * wb[3]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      float x = acc[t];
      x += __shfl_xor(x, 1, 64);
      x += __shfl_xor(x, 2, 64);
      if (q == 0) part[r * 8 + t] = x;
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < TILE; p += blockDim.x) {
    const int l = l0 + p;
    if (l < L) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) v += part[(p + t) * 8 + t];
      out[(long)s * L + l] = apply_tanh ? tanhf(v) : v;
    }
  }
}

__global__ void conv_in1_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                const float* __restrict__ bias, int L, int Cout, int ktaps,
                                float* __restrict__ of, op16_t* __restrict__ op, long ps, int planes, int act,
                                const float* __restrict__ aa, const float* __restrict__ ab, long total) {
  const int pad = (ktaps - 1) / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const long sl = i / Cout;
    const int l = (int)(sl % L);
    const long s = sl / L;
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < ktaps; ++t) {
      const int li = l + t - pad;
      if (li >= 0 && li < L) acc += w[co * ktaps + t] * wav[s * L + li];
    }
    if (of) of[i] = acc;
    if (op) {
      float a = acc;
      if (act == DSN_ACT_ELU) a = dsn_elu(acc);
      else if (act == DSN_ACT_SNAKE) a = dsn_snake(acc, aa[co], ab[co]);
      store_planes(op, ps, planes, i, a);
    }
  }
}

// Cin == 1 convolution, 4 output channels per thread (Cout % 4 == 0, ktaps <= 8): the thread's 4 x ktaps weights
// and biases stay in registers, outputs leave as 16-byte fp32 / 8-byte plane stores (the scalar version above wrote
// 2-byte plane elements)
__global__ void conv_in1_vec_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                    const float* __restrict__ bias, int L, int Cout, int ktaps,
                                    float* __restrict__ of, op16_t* __restrict__ op, long ps, int planes, int act,
                                    const float* __restrict__ aa, const float* __restrict__ ab, long rows) {
  const int cq = Cout >> 2;
  const int c4 = (threadIdx.x % cq) * 4;  // blockDim.x % cq == 0: a thread keeps its channels across rows
  const int pad = (ktaps - 1) / 2;
  float wr[4][8];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) wr[r][t] = t < ktaps ? w[(c4 + r) * ktaps + t] : 0.f;
    if (bias) bv[r] = bias[c4 + r];
  }
  const int rpb = blockDim.x / cq;
  for (long row = blockIdx.x * (long)rpb + threadIdx.x / cq; row < rows; row += (long)gridDim.x * rpb) {
    const int l = (int)(row % L);
    const long s = row / L;
    float x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int li = l + t - pad;
      x[t] = (t < ktaps && li >= 0 && li < L) ? wav[s * L + li] : 0.f;
    }
    f32x4 acc = bv;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[r] += wr[r][t] * x[t];
    const long o = row * Cout + c4;
    if (of) *reinterpret_cast<f32x4*>(of + o) = acc;
    if (op) {
      op16x4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r];
        if (act == DSN_ACT_ELU) v = dsn_elu(v);
        else if (act == DSN_ACT_SNAKE) v = dsn_snake(v, aa[c4 + r], ab[c4 + r]);
        op16_t h, lw;
        dsn_split(v, h, lw, PL_F16(planes));
        hi[r] = h;
        lo[r] = lw;
      }
      *reinterpret_cast<op16x4*>(op + o) = hi;
      if (PL_COUNT(planes) == 2) *reinterpret_cast<op16x4*>(op + ps + o) = lo;
    }
  }
}

__global__ void vae_sample_kernel(const float* __restrict__ enc, const float* __restrict__ noise,
                                  float* __restrict__ y, int D, int T, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long sc = i / T;
    const int c = (int)(sc % D);
    const long s = sc / D;
    const float* row = enc + (s * T + t) * (2L * D);
    const float mean = row[c], scale = row[D + c];
    const float sp = scale > 20.f ? scale : log1pf(expf(scale));
    y[i] = noise[i] * (sp + 1e-4f) + mean;
  }
}

// ------------------------------------------------------------------ RNG
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void randn_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned long long offset) {
  const long n4 = (n + 3) >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long ctr = (unsigned long long)i + offset;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = ((float)(c[r] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float z[4];
    const float r0 = sqrtf(-2.f * logf(u[0])), r1 = sqrtf(-2.f * logf(u[2]));
    const float tw = 6.283185307179586f;
    z[0] = r0 * cosf(tw * u[1]);
    z[1] = r0 * sinf(tw * u[1]);
    z[2] = r1 * cosf(tw * u[3]);
    z[3] = r1 * sinf(tw * u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (i * 4 + r < n) out[i * 4 + r] = z[r];
  }
}

// ------------------------------------------------------------------ weight packing
__global__ void wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale,
                                int R, long inner) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  float s = 0.f;
  for (long i = lane; i < inner; i += 64) {
    const float x = v[(long)row * inner + i];
    s += x * x;
  }
  s = wave_sum(s);
  if (lane == 0) scale[row] = g[row] / sqrtf(s);
}

__global__ void pack_weight_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                   op16_t* __restrict__ dst, long ps, int planes, int mode, int N, int K,
                                   int Cin, int Cout, int kw, int stride, const float* __restrict__ colscale) {
  const long total = (long)N * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float v;
    if (mode == PACK_LINEAR) {
      v = src[(long)n * K + k];
      if (colscale) v *= colscale[k];
    } else if (mode == PACK_LINEAR_SWIGLU) {
      const int F = N / 2, g = n >> 5, w = n & 31;
      const int srow = w < 16 ? 16 * g + w : F + 16 * g + (w - 16);
      v = src[(long)srow * K + k];
      if (colscale) v *= colscale[k];
    } else if (mode == PACK_CONV) {
      const int tap = k / Cin, ci = k - tap * Cin;
      v = src[((long)n * Cin + ci) * kw + tap];
      if (scale) v *= scale[n];
    } else if (mode == PACK_CONV2D) {
      const int cpad = stride;
      const int tap = k / cpad, ci = k - tap * cpad;
      v = (n < Cout && ci < Cin) ? src[((long)n * Cin + ci) * kw + tap] : 0.f;
    } else if (mode == PACK_NIN) {
      v = src[(long)k * N + n];
    } else {  // PACK_CONVT: n = phase*Cout + co ; k = tap*Cin + ci ; kernel index = phase + tap*stride
      const int p = n / Cout, co = n - p * Cout;
      const int tap = k / Cin, ci = k - tap * Cin;
      v = src[((long)ci * Cout + co) * kw + (p + tap * stride)];
      if (scale) v *= scale[ci];
    }
    store_planes(dst, ps, planes, i, v);
  }
}

// one thread per (row, 32-element block): amax -> E8M0 scale -> 32 saturated e4m3 bytes
// colscale (optional): per-k factor multiplied in BEFORE quantisation (LayerNorm gamma folded into the weight)
__global__ void pack_weight_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                       unsigned char* __restrict__ scales, int N, int K, int swiglu,
                                       const float* __restrict__ colscale) {
  const int kb = K >> 5;
  const long total = (long)N * kb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / kb), b = (int)(i - (long)n * kb);
    int srow = n;
    if (swiglu) {
      const int F = N / 2, g = n >> 5, w = n & 31;
      srow = w < 16 ? 16 * g + w : F + 16 * g + (w - 16);
    }
    const f32x4* sp = reinterpret_cast<const f32x4*>(src + (long)srow * K + b * 32);
    f32x4 v[8];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = sp[j];
      if (colscale) v[j] *= *reinterpret_cast<const f32x4*>(colscale + b * 32 + j * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(v[j][r]));
    }
    const int kx = dsn_mx_exp(amax);
    const float inv = dsn_pow2(-kx);
    unsigned* dp = reinterpret_cast<unsigned*>(dst + (long)n * K + b * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) dp[j] = dsn_fp8x4(v[j] * inv);
    scales[i] = (unsigned char)(kx + 127);
  }
}

// colsum[n] = sum_k of the ROUNDED packed operand (plane 0 [+ plane 1]) of row n: the term the folded LayerNorm
// subtracts (mean * colsum) must cancel against exactly the weights the MFMAs multiply.  One wave per row.
__global__ void packed_row_sum_kernel(const op16_t* __restrict__ w, long ps, int planes, int N, int K,
                                      float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63, f16 = PL_F16(planes);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    s += from_op16(w[(long)row * K + k], f16);
    if (PL_COUNT(planes) == 2) s += from_op16(w[ps + (long)row * K + k], f16);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
// The same for an fp8 (MX) packed weight: sum_k of the DEQUANTISED bytes, e4m3 * 2^(scale - 127).  One wave per row;
// lane l sums the dwords l, l + 64, ... of the row (fixed order: every call returns the same bits).
__global__ void fp8_row_sum_kernel(const unsigned char* __restrict__ w, const unsigned char* __restrict__ scales, int N,
                                   int K, float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k4 = lane; k4 < (K >> 2); k4 += 64) {
    const int word = reinterpret_cast<const int*>(w + (long)row * K)[k4];
    const float sc = dsn_pow2((int)scales[(long)row * (K >> 5) + (k4 >> 3)] - 127);
    s += sc * ((__builtin_amdgcn_cvt_f32_fp8(word, 0) + __builtin_amdgcn_cvt_f32_fp8(word, 1)) +
               (__builtin_amdgcn_cvt_f32_fp8(word, 2) + __builtin_amdgcn_cvt_f32_fp8(word, 3)));
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
// out[n] = bias[n] (or 0) + sum_k W[n][k] * beta[k]   (original row order).  One wave per row.
__global__ void bias_plus_wbeta_kernel(const float* __restrict__ W, const float* __restrict__ beta,
                                       const float* __restrict__ bias, int N, int K, float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += W[(long)row * K + k] * beta[k];
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (bias ? bias[row] : 0.f);
}

__global__ void pack_bias_swiglu_kernel(const float* __restrict__ src, float* __restrict__ dst, int N) {
  const int F = N / 2;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const int g = n >> 5, w = n & 31;
    dst[n] = src[w < 16 ? 16 * g + w : F + 16 * g + (w - 16)];
  }
}

__global__ void snake_params_kernel(const float* __restrict__ alpha, const float* __restrict__ beta,
                                    float* __restrict__ a, float* __restrict__ ib, int C) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) {
    a[i] = expf(alpha[i]);
    ib[i] = 1.0f / (expf(beta[i]) + 1e-9f);
  }
}

}  // namespace

// =========================================================================== launchers
void launch_pack_tokens(const float* s0, int C0, const float* s1, int C1, int B, int T, float* df, op16_t* dp,
                        long ps, int planes, hipStream_t st) {
  const long n = (long)B * T * (C0 + C1);
  hipLaunchKernelGGL(pack_tokens_kernel, dim3(grid_for(n)), dim3(TPB), 0, st, s0, C0, s1, C1, B, T, df, dp, ps,
                     planes);
}
void launch_copy_rows(const float* src, float* dst, int rows, int width, long dst_stride, hipStream_t st) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((long)rows * (width / 4))), dim3(TPB), 0, st, src, dst, rows,
                     width / 4, dst_stride);
}
void launch_unpack_tokens(const float* src, float* dst, int B, int C, int T, hipStream_t st) {
  hipLaunchKernelGGL(unpack_tokens_kernel, dim3(grid_for((long)B * C * T)), dim3(TPB), 0, st, src, dst, B, C, T);
}
void launch_to_planes(const float* src, op16_t* dst, long ps, int planes, long n, hipStream_t st) {
  hipLaunchKernelGGL(to_planes_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, st, src, dst, ps, planes, n / 4);
}
void launch_pc_prior(const float* mean, int mean_full, const float* z, float* x, float stdT, int B, int n, int D,
                     int T, hipStream_t st) {
  const long total = (long)B * n * D * T;
  hipLaunchKernelGGL(pc_prior_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, mean, z, x, stdT, n, D, T, total,
                     mean_full);
}
void launch_pc_corrector(float* x, float* xm, const float* sc, const float* z, float step, float gain,
                         const float* norms, float snr, int B, int n, int D, int T, hipStream_t st) {
  const long total = (long)B * n * D * T;
  hipLaunchKernelGGL(pc_corrector_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, x, xm, sc, z, step, gain, norms,
                     B, snr, n, D, T, total);
}
void launch_pc_item_norms(const float* a, long per_item, int B, float* out, hipStream_t st) {
  hipLaunchKernelGGL(pc_item_norm_kernel, dim3(B), dim3(256), 0, st, a, per_item, out);
}
void launch_pc_predictor(float* x, float* xm, const float* y, const float* sc, const float* z, float theta,
                         float dt, float G, float g, int em, int B, int n, int D, int T, hipStream_t st) {
  const long total = (long)B * n * D * T;
  hipLaunchKernelGGL(pc_predictor_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, x, xm, y, sc, z, theta, dt, G, g,
                     em, n, D, T, total);
}
void launch_sigma_mix(const float* y, float* smix, int B, int L, int avg_len, hipStream_t st) {
  const long total = (long)B * L;
  hipLaunchKernelGGL(sigma_mix_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, y, smix, L, avg_len, total);
}
void launch_mix_prior(const float* y, const float* z, float* x, const float* smix, float s1, float s2, int B, int n, int D,
                      int T, hipStream_t st) {
  const long npos = (long)B * D * T;
  hipLaunchKernelGGL(mix_prior_kernel, dim3(grid_for(npos)), dim3(TPB), 0, st, y, z, x, smix, s1, s2, n, D, T, npos);
}
void launch_mix_corrector(float* x, float* xm, const float* sc, const float* z, const float* smix, float sq1, float sq2,
                          float snr, int B, int n, int D, int T, hipStream_t st) {
  const long npos = (long)B * D * T;
  hipLaunchKernelGGL(mix_corrector_kernel, dim3(grid_for(npos)), dim3(TPB), 0, st, x, xm, sc, z, smix, sq1, sq2, snr, n,
                     D, T, npos);
}
void launch_mix_predictor(float* x, float* xm, const float* sc, const float* z, const float* smix, float lambda, float dt,
                          float g, float sqdt, int em, int B, int n, int D, int T, hipStream_t st) {
  const long npos = (long)B * D * T;
  hipLaunchKernelGGL(mix_predictor_kernel, dim3(grid_for(npos)), dim3(TPB), 0, st, x, xm, sc, z, smix, lambda, dt, g,
                     sqdt, em, n, D, T, npos);
}
void launch_sb_update(float* x, const float* est, const float* third, float w_prev, float w_est, float w3, int third_is_y,
                      int B, int n, int D, int T, hipStream_t st) {
  const long total = (long)B * n * D * T;
  hipLaunchKernelGGL(sb_update_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, x, est, third, w_prev, w_est, w3,
                     third_is_y, n, D, T, total);
}
void launch_repeat_sources(const float* y, float* x, int B, int n, int D, int T, hipStream_t st) {
  const long total = (long)B * n * D * T;
  hipLaunchKernelGGL(repeat_sources_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, y, x, n, D, T, total);
}
void launch_residual_norm(float* x, const float* slabs, int nslab, long slab_stride, const float* bias,
                          const float* gamma, const float* beta, op16_t* out, long ps, int planes, int rows, int D,
                          float eps, int do_norm, hipStream_t st, unsigned char* o8s) {
#define RN_LAUNCH(MV, NS_, BT_)                                                                                       \
  hipLaunchKernelGGL((residual_norm_kernel<MV, NS_, BT_>), dim3(grid_for(rows, 4)), dim3(TPB), 0, st, x, slabs, nslab, \
                     slab_stride, bias, gamma, beta, out, ps, planes, rows, D, eps, do_norm, o8s)
#define RN_SWITCH(MV, BT_)                                                                 \
  switch (nslab) {                                                                         \
    case 0: RN_LAUNCH(MV, 0, BT_); break;                                                  \
    case 1: RN_LAUNCH(MV, 1, BT_); break;                                                  \
    case 2: RN_LAUNCH(MV, 2, BT_); break;                                                  \
    case 3: RN_LAUNCH(MV, 3, BT_); break;                                                  \
    case 4: RN_LAUNCH(MV, 4, BT_); break;                                                  \
    case 5: RN_LAUNCH(MV, 5, BT_); break;                                                  \
    case 6: RN_LAUNCH(MV, 6, BT_); break;                                                  \
    case 7: RN_LAUNCH(MV, 7, BT_); break;                                                  \
    default: RN_LAUNCH(MV, 8, BT_); break;  /* pick_ksplit caps the split at 8 */          \
  }
  static const bool no_rowk = getenv("DSN_NO_LN_ROWK") != nullptr;
  if (!no_rowk && D == 4 * TPB && rows > 256 && nslab <= 4) {  // bandwidth-bound: a row per workgroup, everything in flight
#define RNR(NS_)                                                                                                     \
  hipLaunchKernelGGL((residual_norm_row_kernel<NS_>), dim3(rows), dim3(TPB), 0, st, x, slabs, slab_stride, bias, gamma, \
                     beta, out, ps, planes, D, eps, do_norm, o8s)
    switch (nslab) {
      case 0: RNR(0); break;
      case 1: RNR(1); break;
      case 2: RNR(2); break;
      case 3: RNR(3); break;
      default: RNR(4); break;
    }
#undef RNR
    return;
  }
  if (D <= 1024 && rows <= 256) {  // latency-bound: whole-row load batches
    RN_SWITCH(4, true)
  } else if (D <= 1024) {
    RN_SWITCH(4, false)
  } else {
    RN_SWITCH(16, false)
  }
#undef RN_SWITCH
#undef RN_LAUNCH
}
void launch_timestep_features(const float* t, const float* w, int B, int half, op16_t* out, long ps, int planes,
                              hipStream_t st) {
  hipLaunchKernelGGL(timestep_features_kernel, dim3(grid_for((long)B * half)), dim3(TPB), 0, st, t, w, B, half,
                     out, ps, planes);
}
void launch_rope_tables(float* ct, float* stb, int S, int rot, hipStream_t st) {
  hipLaunchKernelGGL(rope_tables_kernel, dim3(grid_for((long)S * rot)), dim3(TPB), 0, st, ct, stb, S, rot);
}
void launch_conv_out1(const op16_t* a, long ps, int planes, const float* w, float* out, int S, int L, int C,
                      int ktaps, int apply_tanh, hipStream_t st) {
  if (ktaps == 7 && C % 32 == 0 && C <= 1024) {
    const size_t smr = ((size_t)7 * C + (size_t)(256 + 6) * 8) * sizeof(float);
    hipLaunchKernelGGL(conv_out1_rows_kernel<7>, dim3(S * ((L + 255) / 256)), dim3(256), smr, st, a, ps, planes, w,
                       out, L, C, apply_tanh);
    return;
  }
  const size_t sm = ((size_t)(64 + ktaps - 1) * (C + 4) + (size_t)ktaps * C + 256) * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};
  if (dsn_first_use_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out1_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int blocks = S * ((L + 63) / 64);
  hipLaunchKernelGGL(conv_out1_kernel, dim3(blocks), dim3(TPB), sm, st, a, ps, planes, w, out, L, C, ktaps,
                     apply_tanh);
}
void launch_conv_in1(const float* wav, const float* w, const float* bias, int S, int L, int Cout, int ktaps,
                     float* of, op16_t* op, long ps, int planes, int act, const float* aa, const float* ab,
                     hipStream_t st) {
  if (Cout % 4 == 0 && TPB % (Cout / 4) == 0 && ktaps <= 8) {
    const long rows = (long)S * L;
    hipLaunchKernelGGL(conv_in1_vec_kernel, dim3(grid_for(rows * (Cout / 4))), dim3(TPB), 0, st, wav, w, bias, L, Cout,
                       ktaps, of, op, ps, planes, act, aa, ab, rows);
    return;
  }
  const long total = (long)S * L * Cout;
  hipLaunchKernelGGL(conv_in1_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, wav, w, bias, L, Cout, ktaps, of, op,
                     ps, planes, act, aa, ab, total);
}
void launch_vae_sample(const float* enc, const float* noise, float* y, int S, int D, int T, hipStream_t st) {
  const long total = (long)S * D * T;
  hipLaunchKernelGGL(vae_sample_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, enc, noise, y, D, T, total);
}
void launch_randn(float* out, long n, unsigned long long seed, unsigned long long offset, hipStream_t st) {
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, st, out, n, seed, offset);
}
void launch_wn_scale(const float* v, const float* g, float* scale, int R, long inner, hipStream_t st) {
  hipLaunchKernelGGL(wn_scale_kernel, dim3(cdiv(R, 4)), dim3(TPB), 0, st, v, g, scale, R, inner);
}
void launch_pack_weight(const float* src, const float* scale, op16_t* dst, long ps, int planes, int mode, int N,
                        int K, int Cin, int Cout, int kw, int stride, hipStream_t st, const float* colscale) {
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for((long)N * K)), dim3(TPB), 0, st, src, scale, dst, ps, planes,
                     mode, N, K, Cin, Cout, kw, stride, colscale);
}
void launch_packed_row_sum(const op16_t* w, long ps, int planes, int N, int K, float* out, hipStream_t st) {
  hipLaunchKernelGGL(packed_row_sum_kernel, dim3(cdiv(N, 4)), dim3(TPB), 0, st, w, ps, planes, N, K, out);
}
void launch_bias_plus_wbeta(const float* W, const float* beta, const float* bias, int N, int K, float* out,
                            hipStream_t st) {
  hipLaunchKernelGGL(bias_plus_wbeta_kernel, dim3(cdiv(N, 4)), dim3(TPB), 0, st, W, beta, bias, N, K, out);
}
void launch_pack_weight_fp8(const float* src, unsigned char* dst, unsigned char* scales, int N, int K, int swiglu,
                            hipStream_t st, const float* colscale) {
  hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3(grid_for((long)N * (K >> 5))), dim3(TPB), 0, st, src, dst, scales, N, K,
                     swiglu, colscale);
}
void launch_fp8_row_sum(const unsigned char* w, const unsigned char* scales, int N, int K, float* out, hipStream_t st) {
  hipLaunchKernelGGL(fp8_row_sum_kernel, dim3(cdiv(N, 4)), dim3(TPB), 0, st, w, scales, N, K, out);
}
void launch_pack_bias_swiglu(const float* src, float* dst, int N, hipStream_t st) {
  hipLaunchKernelGGL(pack_bias_swiglu_kernel, dim3(grid_for(N)), dim3(TPB), 0, st, src, dst, N);
}
void launch_snake_params(const float* alpha, const float* beta, float* a, float* ib, int C, hipStream_t st) {
  hipLaunchKernelGGL(snake_params_kernel, dim3(grid_for(C)), dim3(TPB), 0, st, alpha, beta, a, ib, C);
}

// ------------------------------------------------------------------ SI-SDR dot products
// For every (item, ref source i, est source j): <ref_i, est_j>, and the energies |ref_i|^2, |est_j|^2.
// One workgroup per (item, i, j); wave shuffles + LDS tree; fp32 products accumulated in fp64.
namespace {
__global__ __launch_bounds__(256) void sisdr_dots_kernel(const float* __restrict__ ref, const float* __restrict__ est,
                                                         int n, int L, double* __restrict__ out) {
  __shared__ double red[3][4];
  const int b = blockIdx.x / (n * n);
  const int ij = blockIdx.x - b * n * n;
  const int i = ij / n, j = ij - i * n;
  const float* r = ref + ((long)b * n + i) * L;
  const float* e = est + ((long)b * n + j) * L;
  double sre = 0.0, srr = 0.0, see = 0.0;
  for (int k = threadIdx.x; k < L; k += blockDim.x) {
    const double rv = r[k], ev = e[k];
    sre += rv * ev;
    srr += rv * rv;
    see += ev * ev;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sre += __shfl_xor(sre, o, 64);
    srr += __shfl_xor(srr, o, 64);
    see += __shfl_xor(see, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = sre;
    red[1][wave] = srr;
    red[2][wave] = see;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = out + (long)blockIdx.x * 3;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    o[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  }
}
}  // namespace

void launch_sisdr_dots(const float* ref, const float* est, int B, int n, int L, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sisdr_dots_kernel, dim3(B * n * n), dim3(256), 0, st, ref, est, n, L, out);
}
