// Development probe: what one k-tile of the FF-in row-panel loop costs, ingredient by ingredient, in shader cycles.
// A workgroup of 16 waves (4 per SIMD, one workgroup per CU, 256 workgroups) repeats a synthetic k-tile body:
//   per k-step (2 per k-tile): 4 W-fragment ds_read_b128, then 5 x (1 A-fragment ds_read_b128 + 4 MFMA 16x16x32 f16)
// with the ingredients switched on one at a time:
//   MODE 0  MFMAs only (fragments loaded once, before the loop)
//   MODE 1  + the LDS fragment reads
//   MODE 2  + one s_barrier per k-tile (with the s_waitcnt lgkmcnt(0) in front of it)
//   MODE 3  + the global_load_lds DMA of a (272 + 256) x 64 stage per k-tile (2-stage ring, counted vmcnt as in igemm.hip)
// Prints median cycles per k-tile per wave, cycles per MFMA per SIMD, and the in-kernel clock.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/lab/loop_probe scripts/lab/loop_probe.hip && ./scripts/lab/loop_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int ROWS = 272 + 256, BK = 64, STAGE = ROWS * BK;  // elements (2 bytes)
constexpr int KT = 64;                                        // k-tiles per launch

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int MODE, int MTW>
__global__ __launch_bounds__(1024, 1) void probe(const unsigned short* __restrict__ src, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // fill both stages with something non-trivial (random-ish fp16 around 1)
  for (int i = tid; i < 2 * STAGE; i += 1024) lds[i] = (unsigned short)(0x3800 + ((i * 2654435761u) >> 22));
  __syncthreads();
  f32x4 acc[4][MTW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4, fsw = swz(frow);
  const int a_row_off = (wm * MTW * 16 + frow) * BK;
  const int w_row_off = (272 + wn * 64 + frow) * BK;
  // DMA: 66 row groups of 8 rows dealt over 16 waves (4-5 each); source rows wrap inside a 16 MB buffer
  const int rsub = lane >> 3, cpos = lane & 7;
  const unsigned short* gsrc[5];
#pragma unroll
  for (int gi = 0; gi < 5; ++gi) {
    const int g = wave + gi * 16;
    const int row = g * 8 + rsub;
    gsrc[gi] = src + ((long)(blockIdx.x * 64 + row) % 8192) * 1024 + ((cpos ^ swz(row)) << 3);
  }
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int gi = 0; gi < 5; ++gi) {
      const int g = wave + gi * 16;
      if (g < 66)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[gi] + (kt & 15) * 64),
                                         (__attribute__((address_space(3))) void*)(lds + stage * STAGE + g * 8 * BK), 16, 0, 0);
    }
  };
  u16x8 fw[4], fa[MTW];
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) fw[k] = *reinterpret_cast<const u16x8*>(lds + w_row_off + k * 16 * BK + ((fchunk ^ fsw) << 3));
#pragma unroll
    for (int k = 0; k < MTW; ++k) fa[k] = *reinterpret_cast<const u16x8*>(lds + a_row_off + k * 16 * BK + ((fchunk ^ fsw) << 3));
  }
  if (MODE == 3) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < KT; ++i) {
    if (MODE >= 2) {
      if (MODE == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (MODE == 3 && i + 1 < KT) issue((i + 1) & 1, i + 1);
    }
    const unsigned short* base = lds + (i & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fchunk) ^ fsw) << 3;
      if (MODE >= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fw[k] = *reinterpret_cast<const u16x8*>(base + w_row_off + k * 16 * BK + coff);
      }
#pragma unroll
      for (int tm = 0; tm < MTW; ++tm) {
        if (MODE >= 1) fa[tm] = *reinterpret_cast<const u16x8*>(base + a_row_off + tm * 16 * BK + coff);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw[tn]),
                                                               __builtin_bit_cast(f16x8, fa[MODE >= 1 ? tm : tm]),
                                                               acc[tn][tm], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) s += acc[a][b][0] + acc[a][b][3];
  sink[blockIdx.x * 1024 + tid] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
  }
}

// ---- 8 waves (2 x 4), wave tile 128 x 64 (8 x 4 sub-tiles), tile 256 x 256, BK 64, 2-stage ring ----
//   VAR 0: per k-step 4 W fragments, then the 8 A fragments streamed with a 2-deep request queue
//   VAR 1: quadrant phases (guide's 8-phase shape, one k-tile = 4 phases of 16 MFMAs): B fragments of both column halves
//          and A fragments of one row half live in registers; reads of a phase are issued before its barrier pair;
//          wave row 1 runs one barrier behind wave row 0 (its LDS reads fall under wave row 0's MFMAs and vice versa)
// ---- 8 compute waves (as probe8 VAR 0) + 4 LOADER waves (one per SIMD) that issue every LDS-DMA instruction ----
// compute waves never touch the DMA: per k-tile they wait lgkmcnt(0), barrier, read fragments, MFMA; the loaders wait for
// their own DMA of tile i (vmcnt), join the same barrier, then issue tile i+1 (16 instructions each).
__global__ __launch_bounds__(768, 1) void probe_ld(const unsigned short* __restrict__ src, float* __restrict__ sink,
                                                   unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int R8 = 512, ST8 = R8 * BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * ST8; i += 768) lds[i] = (unsigned short)(0x3800 + ((i * 2654435761u) >> 22));
  __syncthreads();
  if (wave >= 8) {  // ---- loader wave lw = 0..3: row groups lw, lw+4, ... (16 of the 64)
    const int lw = wave - 8;
    const int rsub = lane >> 3, cpos = lane & 7;
    const unsigned short* gsrc[16];
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) {
      const int row = (lw + gi * 4) * 8 + rsub;
      gsrc[gi] = src + ((long)(blockIdx.x * 64 + row) % 8192) * 1024 + ((cpos ^ swz(row)) << 3);
    }
    auto issue = [&](int stage, int kt) {
#pragma unroll
      for (int gi = 0; gi < 16; ++gi)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[gi] + (kt & 15) * 64),
                                         (__attribute__((address_space(3))) void*)(lds + stage * ST8 + (lw + gi * 4) * 8 * BK), 16, 0, 0);
    };
    issue(0, 0);
    for (int i = 0; i < KT; ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 1 < KT) issue((i + 1) & 1, i + 1);
    }
    return;
  }
  const int wm = wave >> 2, wn = wave & 3;
  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4, fsw = swz(frow);
  const int a_row_off = (wm * 128 + frow) * BK;
  const int w_row_off = (256 + wn * 64 + frow) * BK;
  auto ldA = [&](const unsigned short* base, int tm, int ks) {
    return *reinterpret_cast<const u16x8*>(base + a_row_off + tm * 16 * BK + (((ks * 4 + fchunk) ^ fsw) << 3));
  };
  auto ldW = [&](const unsigned short* base, int tn, int ks) {
    return *reinterpret_cast<const u16x8*>(base + w_row_off + tn * 16 * BK + (((ks * 4 + fchunk) ^ fsw) << 3));
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < KT; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned short* base = lds + (i & 1) * ST8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u16x8 fw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) fw[k] = ldW(base, k, ks);
      u16x8 q0 = ldA(base, 0, ks), q1 = ldA(base, 1, ks), q2 = q1;
#pragma unroll
      for (int tm = 0; tm < 8; ++tm) {
        if (tm + 2 < 8) {
          q2 = ldA(base, tm + 2, ks);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw[tn]), __builtin_bit_cast(f16x8, q0),
                                                               acc[tn][tm], 0, 0, 0);
        q0 = q1;
        q1 = q2;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sres = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) sres += acc[a][b][0] + acc[a][b][3];
  sink[blockIdx.x * 1024 + tid] = sres;
  if (lane == 0) {
    stamps[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
  }
}

template <int VAR, int DMA>
__global__ __launch_bounds__(512, 1) void probe8(const unsigned short* __restrict__ src, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int R8 = 512, ST8 = R8 * BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 2 * ST8; i += 512) lds[i] = (unsigned short)(0x3800 + ((i * 2654435761u) >> 22));
  __syncthreads();
  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4, fsw = swz(frow);
  const int a_row_off = (wm * 128 + frow) * BK;
  const int w_row_off = (256 + wn * 64 + frow) * BK;
  const int rsub = lane >> 3, cpos = lane & 7;
  const unsigned short* gsrc[8];
#pragma unroll
  for (int gi = 0; gi < 8; ++gi) {
    const int row = (wave + gi * 8) * 8 + rsub;
    gsrc[gi] = src + ((long)(blockIdx.x * 64 + row) % 8192) * 1024 + ((cpos ^ swz(row)) << 3);
  }
  auto issue_part = [&](int stage, int kt, int g0, int g1) {  // row groups [g0, g1) of this wave's 8
#pragma unroll
    for (int gi = 0; gi < 8; ++gi)
      if (gi >= g0 && gi < g1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[gi] + (kt & 15) * 64),
                                         (__attribute__((address_space(3))) void*)(lds + stage * ST8 + (wave + gi * 8) * 8 * BK), 16, 0, 0);
  };
  if (DMA) issue_part(0, 0, 0, 8);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (VAR == 1 && wm == 1) __builtin_amdgcn_s_barrier();   // stagger: wave row 1 one barrier behind
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  auto ldA = [&](const unsigned short* base, int tm, int ks) {
    return *reinterpret_cast<const u16x8*>(base + a_row_off + tm * 16 * BK + (((ks * 4 + fchunk) ^ fsw) << 3));
  };
  auto ldW = [&](const unsigned short* base, int tn, int ks) {
    return *reinterpret_cast<const u16x8*>(base + w_row_off + tn * 16 * BK + (((ks * 4 + fchunk) ^ fsw) << 3));
  };
  auto mm = [&](const u16x8& w, const u16x8& a, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
  };
  if (VAR == 0) {
    for (int i = 0; i < KT; ++i) {
      if (DMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (DMA && i + 1 < KT) issue_part((i + 1) & 1, i + 1, 0, 8);
      const unsigned short* base = lds + (i & 1) * ST8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u16x8 fw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fw[k] = ldW(base, k, ks);
        u16x8 q0 = ldA(base, 0, ks), q1 = ldA(base, 1, ks), q2 = q1;
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
          if (tm + 2 < 8) {
            q2 = ldA(base, tm + 2, ks);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) mm(fw[tn], q0, acc[tn][tm]);
          q0 = q1;
          q1 = q2;
        }
      }
    }
  } else {
    // phases of k-tile i (wave tile rows 0-63 = row half 0, 64-127 = row half 1; columns 0-31 / 32-63 = column halves):
    //   P0: read B(col half 0) + A(row half 0); MFMA (0,0)   P1: read B(col half 1); MFMA (0,1)
    //   P2: read A(row half 1);                  MFMA (1,1)   P3: --;                  MFMA (1,0)
    // every phase: [reads, a quarter of the next tile's DMA] barrier, lgkmcnt(0), 16 MFMAs, barrier
    u16x8 fb[2][2][2];  // [col half][tn in half][ks]
    u16x8 fa[4][2];     // [tm in row half][ks]
    for (int i = 0; i < KT; ++i) {
      const unsigned short* base = lds + (i & 1) * ST8;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int rh = ph >> 1, ch = (ph == 1 || ph == 2) ? 1 : 0;
        if (ph == 0 || ph == 1) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[ch][t][ks] = ldW(base, ch * 2 + t, ks);
        }
        if (ph == 0 || ph == 2) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[t][ks] = ldA(base, rh * 4 + t, ks);
        }
        if (DMA && i + 1 < KT) issue_part((i + 1) & 1, i + 1, ph * 2, ph * 2 + 2);
        if (ph == 3) {  // the next tile must have landed before its first reads (phase 0 of i+1)
          if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) mm(fb[ch][u][ks], fa[t][ks], acc[ch * 2 + u][rh * 4 + t]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  if (VAR == 1 && wm == 0) __builtin_amdgcn_s_barrier();   // pair the staggered wave row's extra barrier
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sres = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) sres += acc[a][b][0] + acc[a][b][3];
  sink[blockIdx.x * 1024 + tid] = sres;
  if (lane == 0) {
    stamps[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
  }
}

template <class K>
static void run8k(const char* name, K k, int threads, const unsigned short* src, float* sink, unsigned long long* stamps);
template <int VAR, int DMA>
static void run8(const char* name, const unsigned short* src, float* sink, unsigned long long* stamps) {
  run8k(name, probe8<VAR, DMA>, 512, src, sink, stamps);
}
template <class K>
static void run8k(const char* name, K k, int threads, const unsigned short* src, float* sink, unsigned long long* stamps) {
  const size_t smem = 2 * 512 * BK * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 60; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(threads), smem, 0, src, sink, stamps);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(threads), smem, 0, src, sink, stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 16 * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, rt;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) {
      cyc.push_back((double)h[2 * (b * 16 + w)]);
      rt.push_back((double)h[2 * (b * 16 + w) + 1]);
    }
  std::sort(cyc.begin(), cyc.end());
  std::sort(rt.begin(), rt.end());
  const double c = cyc[cyc.size() / 2], r = rt[rt.size() / 2];
  const double per_kt = c / KT;
  const double mfma_per_simd = 2.0 * 64;  // 2 waves per SIMD x 64 MFMAs per wave and k-tile
  const double flops = 256.0 * 8 * KT * 64 * 2.0 * 16 * 16 * 32;
  printf("%-44s 8 waves  %7.0f cycles per k-tile per wave  = %5.1f cycles per MFMA per SIMD   clock %.2f GHz   %6.1f us per launch  "
         "%6.0f TFLOP/s\n", name, per_kt, per_kt / mfma_per_simd, c / r * 0.1, ms / 20 * 1e3, flops / (ms / 20 * 1e-3) / 1e12);
}

template <int MODE, int MTW>
static void run(const char* name, const unsigned short* src, float* sink, unsigned long long* stamps) {
  auto k = probe<MODE, MTW>;
  const size_t smem = 2 * STAGE * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 60; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(1024), smem, 0, src, sink, stamps);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(1024), smem, 0, src, sink, stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 16 * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, rt;
  for (int i = 0; i < 256 * 16; ++i) {
    cyc.push_back((double)h[2 * i]);
    rt.push_back((double)h[2 * i + 1]);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(rt.begin(), rt.end());
  const double c = cyc[cyc.size() / 2], r = rt[rt.size() / 2];
  const double per_kt = c / KT;
  const double mfma_per_simd = 4.0 * MTW * 4 * 2;  // 4 waves per SIMD
  const double flops = 256.0 * 16 * KT * MTW * 4 * 2 * 2.0 * 16 * 16 * 32;
  printf("%-44s MTW=%d  %7.0f cycles per k-tile per wave  = %5.1f cycles per MFMA per SIMD   clock %.2f GHz   %6.1f us per launch  "
         "%6.0f TFLOP/s\n", name, MTW, per_kt, per_kt / mfma_per_simd, c / r * 0.1, ms / 20 * 1e3, flops / (ms / 20 * 1e-3) / 1e12);
}

int main() {
  unsigned short* src;
  float* sink;
  unsigned long long* stamps;
  hipMalloc(&src, 8192L * 1024 * 2 + 4096);
  hipMemset(src, 0x3c, 8192L * 1024 * 2 + 4096);
  hipMalloc(&sink, 256 * 1024 * 4);
  hipMalloc(&stamps, 256 * 16 * 2 * 8);
  for (int rnd = 0; rnd < 2; ++rnd) {
    run<0, 5>("MFMAs only (operands in registers)", src, sink, stamps);
    run<1, 5>("+ LDS fragment reads", src, sink, stamps);
    run<2, 5>("+ barrier per k-tile", src, sink, stamps);
    run<3, 5>("+ LDS-DMA of the stage (2-stage ring)", src, sink, stamps);
    run<1, 4>("+ LDS fragment reads", src, sink, stamps);
    run<3, 4>("+ LDS-DMA of the stage (2-stage ring)", src, sink, stamps);
    run8<0, 0>("8w 128x64 tiles, A queue, no DMA", src, sink, stamps);
    run8<0, 1>("8w 128x64 tiles, A queue, + DMA", src, sink, stamps);
    run8<1, 0>("8w quadrant phases + stagger, no DMA", src, sink, stamps);
    run8<1, 1>("8w quadrant phases + stagger, + DMA", src, sink, stamps);
    run8k("8 compute + 4 loader waves, DMA by the loaders", probe_ld, 768, src, sink, stamps);
  }
  return 0;
}
