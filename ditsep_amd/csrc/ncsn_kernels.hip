// Non-GEMM kernels of the NCSN++ latent score network (channels-last "image" activations:
// row = (item, y, x), y = latent channel axis (64), x = latent frame axis).
// GroupNorm statistics / apply(+SiLU), separable FIR [1,3,3,1] 2x resampling (the reference's
// upfirdn2d CUDA op, src/models/diffsep/ncsnpp_utils/op/upfirdn2d_kernel.cu:49-207, in closed
// form), Gaussian-Fourier time features, input packing and the final 1x1 output layer.
// All HBM-bound byte movers: 16-byte channel-contiguous accesses.
#include "kernels.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;
inline int grid_for(long n, int per_block = TPB, int cap = 256 * 16) {
  long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ void store_planes4(op16_t* dst, long ps, int planes, long i, const f32x4& v) {
  op16x4 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    op16_t h, l;
    dsn_split(v[r], h, l, PL_F16(planes));
    hi[r] = h;
    lo[r] = l;
  }
  *reinterpret_cast<op16x4*>(dst + i) = hi;
  if (PL_COUNT(planes) == 2) *reinterpret_cast<op16x4*>(dst + ps + i) = lo;
}

// xt [B][n][H][T], mix [B][1][H][T]  ->  NHWC [B][H][Wp][Cp], zero in the channel / frame padding
__global__ void ncsn_pack_kernel(const float* __restrict__ xt, const float* __restrict__ mix, int n, int H, int T,
                                 int Wp, int Cp, float* __restrict__ of, op16_t* __restrict__ op, long ps, int planes,
                                 long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    long r = i / Cp;
    const int x = (int)(r % Wp);
    r /= Wp;
    const int y = (int)(r % H);
    const long b = r / H;
    float v = 0.f;
    if (x < T) {
      if (c < n) v = xt[((b * n + c) * H + y) * T + x];
      else if (c == n) v = mix[(b * H + y) * T + x];
    }
    if (of) of[i] = v;
    if (op) {
      op16_t h, l;
      dsn_split(v, h, l, PL_F16(planes));
      op[i] = h;
      if (PL_COUNT(planes) == 2) op[ps + i] = l;
    }
  }
}

// GroupNorm statistics as slice partials: stats[B][S][C/4][2] = (mean, M2) of the (<= 64 rows) x 4 channels of slice
// s = rows [64 s, 64 s + 64) of the item, quad q -- the layout the GEMM epilogue writes for tensors it produces
// (GemmDesc::gn_stats).  One wave per (slice, 16 quads): lane (quad ql, row part rp) keeps its 16 rows x 4 channels in
// registers (one pass over memory), the four row parts meet through a fixed xor tree; plain stores.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, long bstride, int rstride, int C,
                                                       int HW, float* __restrict__ stats) {
  const int nq = C >> 2;
  const int nqb = (nq + 15) >> 4;
  const int S = (HW + 63) >> 6;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, ql = lane & 15, rp = lane >> 4;
  const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= S * nqb) return;  // whole waves leave together
  const int sl = item / nqb, q = (item - sl * nqb) * 16 + ql;
  const int r0 = sl * 64 + rp * 16, r1 = min(sl * 64 + 64, HW);
  const bool live = q < nq;
  const float* xp = x + (long)b * bstride + (live ? q : 0) * 4;
  f32x4 v[16];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const bool ok = live && r0 + k < r1;
    v[k] = ok ? *reinterpret_cast<const f32x4*>(xp + (long)(r0 + k) * rstride) : f32x4{0.f, 0.f, 0.f, 0.f};
    sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float cnt = (float)((r1 - sl * 64) * 4);
  const float mean = sum / cnt;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (r0 + k < r1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) m2 += (v[k][c] - mean) * (v[k][c] - mean);
    }
  }
  m2 += __shfl_xor(m2, 16, 64);
  m2 += __shfl_xor(m2, 32, 64);
  if (live && rp == 0) *reinterpret_cast<float2*>(stats + (((long)b * S + sl) * nq + q) * 2) = float2{mean, m2};
}

// y = (x - mean) * rstd * gamma + beta [, SiLU]  ->  contiguous [B][HW][C] fp32 and/or planes.
// grid (row chunks, B): a block first combines the slice partials of its item's G groups (fixed order, Chan's
// parallel formula: M2 = sum M2_p + sum n_p (mean_p - mean)^2) into LDS, then streams its rows.
__global__ void gn_apply_kernel(const float* __restrict__ x, long bstride, int rstride, int C, int G, int HW,
                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int silu, float* __restrict__ of,
                                op16_t* __restrict__ op, long ps, int planes, int rows_per_block) {
  __shared__ float gm[64], gr[64];
  const int nq = C >> 2;
  const int cpg = C / G, qpg = cpg >> 2;
  const int S = (HW + 63) >> 6;
  const int b = blockIdx.y;
  {
    // statistics prologue: TPG lanes per group share the (slice, quad) partials; lane-local sums in index order, then a
    // fixed xor tree -- the same bits on every launch.  Two passes (mean, then M2 about it: Chan's combine).
    int tpg = 1;
    while (tpg < 64 && G * tpg * 2 <= (int)blockDim.x) tpg *= 2;
    const int g = threadIdx.x / tpg, sub = threadIdx.x - g * tpg;
    const bool live = g < G;
    const float2* sp = reinterpret_cast<const float2*>(stats) + (long)b * S * nq + (live ? g : 0) * qpg;
    const int items = S * qpg;
    float wsum = 0.f;
    if (live)
      for (int it = sub; it < items; it += tpg) {
        const int sl = it / qpg, q = it - sl * qpg;
        const float cnt = (float)((min(sl * 64 + 64, HW) - sl * 64) * 4);
        wsum += cnt * sp[(long)sl * nq + q].x;
      }
    for (int o = tpg >> 1; o >= 1; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
    const float ntot = (float)HW * (float)cpg;
    const float mean = wsum / ntot;
    float m2 = 0.f;
    if (live)
      for (int it = sub; it < items; it += tpg) {
        const int sl = it / qpg, q = it - sl * qpg;
        const float cnt = (float)((min(sl * 64 + 64, HW) - sl * 64) * 4);
        const float2 pr = sp[(long)sl * nq + q];
        const float dm = pr.x - mean;
        m2 += pr.y + cnt * dm * dm;
      }
    for (int o = tpg >> 1; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o, 64);
    if (live && sub == 0) {
      gm[g] = mean;
      gr[g] = rsqrtf(m2 / ntot + eps);
    }
  }
  __syncthreads();
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(r_begin + rows_per_block, HW);
  const int total4 = (r_end - r_begin) * nq;
  const float* xb = x + (long)b * bstride;
  constexpr int UN = 4;  // independent loads in flight per thread (one per iteration left the kernel latency-bound)
  for (int i0 = threadIdx.x; i0 < total4; i0 += UN * blockDim.x) {
    f32x4 v[UN];
    int q[UN], r[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * blockDim.x;
      const bool ok = i < total4;
      q[u] = ok ? i % nq : 0;
      r[u] = ok ? r_begin + i / nq : r_begin;
      v[u] = *reinterpret_cast<const f32x4*>(xb + (long)r[u] * rstride + q[u] * 4);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (i0 + u * (int)blockDim.x >= total4) break;
      const int g = (q[u] * 4) / cpg;
      const float mean = gm[g], rstd = gr[g];
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + q[u] * 4);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + q[u] * 4);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = (v[u][k] - mean) * rstd * ga[k] + be[k];
        o[k] = silu ? dsn_silu(t) : t;
      }
      const long oi = ((long)b * HW + r[u]) * nq + q[u];
      if (of) reinterpret_cast<f32x4*>(of)[oi] = o;
      if (op) store_planes4(op, ps, planes, oi * 4, o);
    }
  }
}

// 2x FIR resampling with the separable [1,3,3,1] filter on channels-last images (closed form of
// upfirdn2d: up  out[2j] = .25 x[j-1] + .75 x[j], out[2j+1] = .75 x[j] + .25 x[j+1];
//            down out[o] = .125 x[2o-1] + .375 x[2o] + .375 x[2o+1] + .125 x[2o+2]; zeros outside)
__global__ void fir2d_kernel(const float* __restrict__ x, long bstride, int rstride, int C, int H, int W, int up,
                             const float* __restrict__ add, float* __restrict__ of, op16_t* __restrict__ op, long ps,
                             int planes, long total4) {
  const int nq = C >> 2;
  const int Ho = up ? 2 * H : H / 2, Wo = up ? 2 * W : W / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % nq);
    long r = i / nq;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const long b = r / Ho;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xb = x + b * bstride + q * 4;
    if (up) {
      const int jy = yo >> 1, jx = xo >> 1;
      const int y0 = (yo & 1) ? jy : jy - 1, x0 = (xo & 1) ? jx : jx - 1;
      const float wy0 = (yo & 1) ? 0.75f : 0.25f, wx0 = (xo & 1) ? 0.75f : 0.25f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = y0 + a;
        if ((unsigned)yy >= (unsigned)H) continue;
        const float wy = a == 0 ? wy0 : 1.f - wy0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int xx = x0 + c;
          if ((unsigned)xx >= (unsigned)W) continue;
          const float w = wy * (c == 0 ? wx0 : 1.f - wx0);
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long)yy * W + xx) * rstride);
          acc += v * w;
        }
      }
    } else {
      const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int yy = 2 * yo - 1 + a;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xx = 2 * xo - 1 + c;
          if ((unsigned)xx >= (unsigned)W) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long)yy * W + xx) * rstride);
          acc += v * (k[a] * k[c]);
        }
      }
    }
    if (add) acc += reinterpret_cast<const f32x4*>(add)[i];
    if (of) reinterpret_cast<f32x4*>(of)[i] = acc;
    if (op) store_planes4(op, ps, planes, i * 4, acc);
  }
}

// GaussianFourierProjection(log t): [sin(2 pi log(t) W), cos(...)] -> planes [B][2*nf]
__global__ void ncsn_fourier_kernel(const float* __restrict__ t, const float* __restrict__ w, int B, int nf,
                                    op16_t* __restrict__ out, long ps, int planes) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * nf; i += gridDim.x * blockDim.x) {
    const int b = i / nf, k = i - b * nf;
    const float p = logf(t[b]) * w[k] * 2.f * 3.14159265358979323846f;
    op16_t h, l;
    dsn_split(sinf(p), h, l, PL_F16(planes));
    out[(long)b * 2 * nf + k] = h;
    if (PL_COUNT(planes) == 2) out[ps + (long)b * 2 * nf + k] = l;
    dsn_split(cosf(p), h, l, PL_F16(planes));
    out[(long)b * 2 * nf + nf + k] = h;
    if (PL_COUNT(planes) == 2) out[ps + (long)b * 2 * nf + nf + k] = l;
  }
}

// h = pyramid / t ; score = output_layer(h) (1x1, cin -> n) written token-major [(b*T + x)][s*H + y]
__global__ void ncsn_output_kernel(const float* __restrict__ pyr, int Cp, const float* __restrict__ t,
                                   const float* __restrict__ w, const float* __restrict__ bias, int cin, int n, int H,
                                   int T, int Wp, float* __restrict__ score, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i % H);
    long r = i / H;
    const int s = (int)(r % n);
    r /= n;
    const int x = (int)(r % T);
    const long b = r / T;
    const float* p = pyr + ((b * H + y) * Wp + x) * Cp;
    const float it = 1.f / t[b];
    float acc = bias[s];
    for (int c = 0; c < cin; ++c) acc += w[s * cin + c] * (p[c] * it);
    score[i] = acc;
  }
}

}  // namespace

void launch_ncsn_pack(const float* xt, const float* mix, int B, int n, int H, int T, int Wp, int Cp, float* of,
                      op16_t* op, long ps, int planes, hipStream_t st) {
  const long total = (long)B * H * Wp * Cp;
  hipLaunchKernelGGL(ncsn_pack_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, xt, mix, n, H, T, Wp, Cp, of, op, ps,
                     planes, total);
}
void launch_gn_stats(const float* x, long bstride, int rstride, int C, int G, int B, int HW, float* stats,
                     hipStream_t st) {
  if (G > 64 || C % (4 * G) != 0) return;  // unsupported shape: caller validates (engine: C <= 1024)
  const int S = (HW + 63) / 64;
  const long items = (long)S * cdiv(C / 4, 16);  // one wave each
  hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(items, 4), B), dim3(256), 0, st, x, bstride, rstride, C, HW, stats);
}
void launch_gn_apply(const float* x, long bstride, int rstride, int C, int G, int B, int HW, const float* stats,
                     const float* gamma, const float* beta, float eps, int silu, float* of, op16_t* op, long ps,
                     int planes, hipStream_t st) {
  // enough blocks to fill the chip, each with at least ~4k elements behind its statistics prologue
  const int rows_per_block = std::max(cdiv(4096, C), cdiv((long)HW * B, 2048));
  hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(HW, rows_per_block), B), dim3(TPB), 0, st, x, bstride, rstride, C, G, HW,
                     stats, gamma, beta, eps, silu, of, op, ps, planes, rows_per_block);
}
void launch_fir2d(const float* x, long bstride, int rstride, int C, int B, int H, int W, int up, const float* add,
                  float* of, op16_t* op, long ps, int planes, hipStream_t st) {
  const int Ho = up ? 2 * H : H / 2, Wo = up ? 2 * W : W / 2;
  const long total4 = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(fir2d_kernel, dim3(grid_for(total4)), dim3(TPB), 0, st, x, bstride, rstride, C, H, W, up, add, of,
                     op, ps, planes, total4);
}
void launch_ncsn_fourier(const float* t, const float* w, int B, int nf, op16_t* out, long ps, int planes,
                         hipStream_t st) {
  hipLaunchKernelGGL(ncsn_fourier_kernel, dim3(grid_for((long)B * nf)), dim3(TPB), 0, st, t, w, B, nf, out, ps, planes);
}
void launch_ncsn_output(const float* pyr, int Cp, const float* t, const float* w, const float* bias, int cin, int n,
                        int B, int H, int T, int Wp, float* score, hipStream_t st) {
  const long total = (long)B * T * n * H;
  hipLaunchKernelGGL(ncsn_output_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, pyr, Cp, t, w, bias, cin, n, H, T, Wp,
                     score, total);
}
