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
  }
  return 0;
}
