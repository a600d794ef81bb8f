// Development lab for the DiT layer GEMMs (M = 2112 token rows): standalone timing of main-loop structures and
// ablations (staging only / compute only) on the real shapes, weights rotated over many buffers so that they come
// from HBM as in the pipeline (24 layers x 33 MB of weights never stay cached).  Not part of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/lab/gemm_lab scripts/lab/gemm_lab.hip
//   ./gemm_lab                 (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

typedef unsigned short op16_t;
typedef __attribute__((ext_vector_type(8))) unsigned short op16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHK(x)                                                                            \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

struct Prob {
  const op16_t* A;  // [M][K]
  const op16_t* W;  // [N][K]
  op16_t* C;        // [M][N] fp16 (or [M][N/2] with swiglu)
  int M, N, K;
  int panel_rows, tiles_m, tiles_n;
  int m_fast;
  const op16_t* Wnext;  // weights of the NEXT launch: touched (one dword per 128-B line) so they sit in the memory-side cache
  long wnext_bytes;
  int pf_mode;          // 0 off, 1 plain loads, 2 nontemporal loads
  unsigned long long* stamps;  // [grid][4]: s_memtime / s_memrealtime at kernel start and end (wave 0)
};

template <int TBK>
__device__ __forceinline__ int swzk(int row) {
  return TBK == 32 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7);
}

__device__ __forceinline__ f32x4 mfma(const op16x8& w, const op16x8& a, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
}

// MODE 0 full, 1 staging only (no LDS reads / MFMA), 2 compute only (no in-loop staging)
// WM x WN waves; MT row sub-tiles per panel dealt over the WM wave rows; every wave owns NTW column sub-tiles.
template <int WM, int WN, int MT, int NTW, int TBK, int NST, int MODE, int SWIGLU>
__global__ __launch_bounds__(WM* WN * 64, 1) void lab_kernel(const Prob d, const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];
  constexpr int NWAVES = WM * WN;
  constexpr int MTW = (MT + WM - 1) / WM;
  constexpr int TBN = WN * NTW * 16;
  constexpr int AROWS = MT * 16;
  constexpr int ROWS = AROWS + TBN;
  constexpr int STAGE_ELEMS = ROWS * TBK;
  constexpr int CPR = TBK / 8;
  constexpr int RPG = 64 / CPR;
  constexpr int GROUPS = ROWS / RPG;
  static_assert(ROWS % RPG == 0, "rows must fill whole glds groups");
  constexpr int GPW = (GROUPS + NWAVES - 1) / NWAVES;
  constexpr int REM = GROUPS % NWAVES;
  constexpr int KS = TBK / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave - wave_m * WN;
  constexpr int MBASE = MT / WM, MREM = MT % WM;
  const int my_mt = MBASE + (wave_m < MREM ? 1 : 0);
  const int my_row0 = 16 * (wave_m * MBASE + min(wave_m, MREM));
  const int my_groups = (REM == 0 || wave < REM) ? GPW : GPW - 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tile_m = d.m_fast ? tile % d.tiles_m : tile / d.tiles_n;
  const int tile_n = d.m_fast ? tile / d.tiles_m : tile - tile_m * d.tiles_n;
  const int m0 = tile_m * d.panel_rows, n0 = tile_n * TBN;
  const int m_end = min(m0 + d.panel_rows, d.M);
  const int nkt = d.K / TBK;

  unsigned long long t0c = 0, t0r = 0;
  if (d.stamps && tid == 0) {
    t0c = __builtin_amdgcn_s_memtime();
    t0r = __builtin_amdgcn_s_memrealtime();
  }
  if (d.pf_mode) {  // prefetch this workgroup's slice of the next launch's weights
    const long lines = d.wnext_bytes / 128;
    const long per = (lines + gridDim.x - 1) / gridDim.x;
    const long l0 = (long)blockIdx.x * per;
    for (long l = l0 + tid; l < min(l0 + per, lines); l += blockDim.x) {
      const unsigned* src = reinterpret_cast<const unsigned*>(d.Wnext) + l * 32;
      unsigned v;
      if (d.pf_mode == 2) v = __builtin_nontemporal_load(src);
      else v = *reinterpret_cast<const volatile unsigned*>(src);
      asm volatile("" ::"v"(v));
    }
  }
  const int rsub = lane / CPR, cpos = lane % CPR;
  const op16_t* rptr[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + gi * NWAVES;
    const bool is_a = g < AROWS / RPG;
    const int row = (is_a ? g : g - AROWS / RPG) * RPG + rsub;
    const int gchunk = cpos ^ swzk<TBK>(row);
    const int idx = (is_a ? m0 : n0) + row;
    const bool ok = g < GROUPS && (is_a ? idx < m_end : idx < d.N);
    rptr[gi] = ok ? (is_a ? d.A : d.W) + (long)idx * d.K + gchunk * 8 : nullptr;
  }
  const op16_t* zsrc = zero_page + cpos * 8;

  auto issue = [&](int stage, int kt) {
    op16_t* sbase = lds + stage * STAGE_ELEMS;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (gi < my_groups) {
        const int g = wave + gi * NWAVES;
        const op16_t* gp = rptr[gi] ? rptr[gi] + kt * TBK : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(sbase + g * RPG * TBK), 16, 0, 0);
      }
    }
  };

  f32x4 acc[NTW][MTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fchunk = lane >> 4;
  const int fsw = swzk<TBK>(frow);
  const int a_row_off = (my_row0 + frow) * TBK;
  const int w_row_off = (AROWS + wave_n * NTW * 16 + frow) * TBK;

#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2)
    if (s2 < nkt) issue(s2, s2);

  for (int i = 0; i < nkt; ++i) {
    const int younger = min(NST - 2, nkt - 1 - i);
    if (MODE == 2) {
      if (i == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else if (NST >= 3 && younger >= 1) {
      // leave `younger` tiles in flight (counted wait: younger x this wave's loads per tile)
      const int mine = (REM == 0 || wave < REM) ? GPW : GPW - 1;
#define WAITY(Y)                                                                                   \
  if (younger == Y) {                                                                              \
    if (mine == GPW)                                                                               \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((Y * GPW) > 63 ? 63 : (Y * GPW)) : "memory");        \
    else                                                                                           \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((Y * (GPW - 1)) > 63 ? 63 : (Y * (GPW - 1))) : "memory"); \
  }
      WAITY(1) WAITY(2) WAITY(3) WAITY(4) WAITY(5) WAITY(6)
#undef WAITY
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (MODE != 2 && i + NST - 1 < nkt) issue((i + NST - 1) % NST, i + NST - 1);
    if (MODE == 1) continue;

    const op16_t* base = lds + (MODE == 2 ? 0 : (i % NST)) * STAGE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = ((ks * 4 + fchunk) ^ fsw) * 8;
      op16x8 fw[NTW];
#pragma unroll
      for (int k = 0; k < NTW; ++k) fw[k] = *reinterpret_cast<const op16x8*>(base + w_row_off + k * 16 * TBK + coff);
#pragma unroll
      for (int tm = 0; tm < MTW; ++tm) {
        if (tm < my_mt) {
          const op16x8 fa = *reinterpret_cast<const op16x8*>(base + a_row_off + tm * 16 * TBK + coff);
#pragma unroll
          for (int tn = 0; tn < NTW; ++tn) acc[tn][tm] = mfma(fw[tn], fa, acc[tn][tm]);
        }
      }
    }
  }
  if (d.stamps && tid == 0) {
    d.stamps[blockIdx.x * 4 + 0] = t0c;
    d.stamps[blockIdx.x * 4 + 1] = t0r;
    d.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
    d.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
  // epilogue: lane owns 4 consecutive channels (n) of row m = .. + (lane & 15)
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int tm = 0; tm < MTW; ++tm) {
    const int m = m0 + my_row0 + tm * 16 + (lane & 15);
    if (tm >= my_mt || m >= m_end) continue;
    if (SWIGLU) {
#pragma unroll
      for (int tn = 0; tn < NTW; tn += 2) {
        const int n = n0 + wave_n * NTW * 16 + tn * 16 + nq;  // packed: 16 value cols then 16 gate cols
        if (n >= d.N) continue;
        const f32x4 v = acc[tn][tm], g = acc[tn + 1][tm];
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(unsigned short, (_Float16)(v[r] * (g[r] / (1.f + __expf(-g[r])))));
        const int no = (n0 + wave_n * NTW * 16 + tn * 16) / 2 + nq;
        *reinterpret_cast<uint2*>(d.C + (long)m * (d.N / 2) + no) =
            uint2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < NTW; ++tn) {
        const int n = n0 + wave_n * NTW * 16 + tn * 16 + nq;
        if (n >= d.N) continue;
        const f32x4 v = acc[tn][tm];
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(unsigned short, (_Float16)v[r]);
        *reinterpret_cast<uint2*>(d.C + (long)m * d.N + n) =
            uint2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong structure: 8 waves = 2 wave rows x 4 wave columns; the wave rows are the two SIMD partners (waves w and
// w + 4 share a SIMD) and alternate roles every barrier interval: one group issues its whole k-tile of MFMAs from
// fragments already in registers while the other fetches its next k-tile's fragments from LDS (and everyone issues
// its share of the global->LDS prefetch), so the matrix pipe of every SIMD always has a wave feeding it and no MFMA
// waits on an LDS read.  BK = 32 k-tiles in a ring of NST slots, NST - 1 tiles in flight behind counted vmcnt.
//   interval 2T   : group 0 MFMA(T)              | group 1 reads fragments(T)
//   interval 2T+1 : group 0 reads fragments(T+1) | group 1 MFMA(T)            (all waves: issue tile T + NST)
// Tile T is read from LDS in intervals 2T-1 (group 0) and 2T (group 1): landed by barrier 2T-1 (every wave waits for
// its own loads of tile T+1 before barrier 2T+1), slot refilled after barrier 2T+1.
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int NST, int SWIGLU>
__global__ __launch_bounds__(512, 1) void pp_kernel(const Prob d, const op16_t* __restrict__ zero_page) {
  extern __shared__ __attribute__((aligned(16))) op16_t lds[];
  constexpr int TBK = 32, NWAVES = 8, NTW = 4;
  constexpr int MTW = (MT + 1) / 2;
  constexpr int TBN = 256;
  constexpr int AROWS = MT * 16;
  constexpr int ROWS = AROWS + TBN;
  constexpr int STAGE_ELEMS = ROWS * TBK;
  constexpr int GROUPS = ROWS / 16;  // 16 rows (of 64 B) per glds wave-instruction
  constexpr int GPW = (GROUPS + NWAVES - 1) / NWAVES;
  constexpr int REM = GROUPS % NWAVES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wave_n = wave & 3;
  const int my_mt = grp == 0 ? MTW : MT - MTW;
  const int my_row0 = grp == 0 ? 0 : MTW * 16;
  const bool full = (REM == 0 || wave < REM);
  const int my_groups = full ? GPW : GPW - 1;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  const int tile_m = d.m_fast ? tile % d.tiles_m : tile / d.tiles_n;
  const int tile_n = d.m_fast ? tile / d.tiles_m : tile - tile_m * d.tiles_n;
  const int m0 = tile_m * d.panel_rows, n0 = tile_n * TBN;
  const int m_end = min(m0 + d.panel_rows, d.M);
  const int nkt = d.K / TBK;

  unsigned long long t0c = 0, t0r = 0;
  if (d.stamps && tid == 0) {
    t0c = __builtin_amdgcn_s_memtime();
    t0r = __builtin_amdgcn_s_memrealtime();
  }
  const int rsub = lane >> 2, cpos = lane & 3;
  const op16_t* rptr[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave + gi * NWAVES;
    const bool is_a = g < AROWS / 16;
    const int row = (is_a ? g : g - AROWS / 16) * 16 + rsub;
    const int gchunk = cpos ^ swzk<TBK>(row);
    const int idx = (is_a ? m0 : n0) + row;
    const bool ok = g < GROUPS && (is_a ? idx < m_end : idx < d.N);
    rptr[gi] = ok ? (is_a ? d.A : d.W) + (long)idx * d.K + gchunk * 8 : nullptr;
  }
  const op16_t* zsrc = zero_page + cpos * 8;
  auto issue = [&](int kt) {
    op16_t* sbase = lds + (kt % NST) * STAGE_ELEMS;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (gi < my_groups) {
        const int g = wave + gi * NWAVES;
        const op16_t* gp = rptr[gi] ? rptr[gi] + kt * TBK : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(sbase + g * 16 * TBK), 16, 0, 0);
      }
    }
  };
  // wait until all but the `younger` most recent tiles' loads of this wave have landed
  auto wait_younger = [&](int younger) {
#define WY(Y)                                                                               \
  if (younger == Y) {                                                                       \
    if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * GPW) : "memory");               \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * (GPW - 1)) : "memory");              \
  }
    WY(0) WY(1) WY(2) WY(3) WY(4)
#undef WY
  };

  f32x4 acc[NTW][MTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < MTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  const int coff = (fchunk ^ swzk<TBK>(frow)) * 8;
  const int a_off = (my_row0 + frow) * TBK + coff;
  const int w_off = (AROWS + wave_n * 64 + frow) * TBK + coff;
  op16x8 fa[MTW], fw[NTW];
  auto load_frags = [&](int kt) {
    const op16_t* base = lds + (kt % NST) * STAGE_ELEMS;
#pragma unroll
    for (int k = 0; k < NTW; ++k) fw[k] = *reinterpret_cast<const op16x8*>(base + w_off + k * 16 * TBK);
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm)
      if (tm < my_mt) fa[tm] = *reinterpret_cast<const op16x8*>(base + a_off + tm * 16 * TBK);
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm)
      if (tm < my_mt) {
#pragma unroll
        for (int tn = 0; tn < NTW; ++tn) acc[tn][tm] = mfma(fw[tn], fa[tm], acc[tn][tm]);
      }
  };

  const int npre = min(NST, nkt);
  for (int s2 = 0; s2 < npre; ++s2) issue(s2);
  wait_younger(npre - 1);
  __builtin_amdgcn_s_barrier();  // barrier -1: tile 0 visible
  if (grp == 0) {
    load_frags(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // interval j: group (j & 1) issues MFMAs of tile (j - grp) / 2, the other reads the fragments of its next tile
  for (int j = 0; j < 2 * nkt; ++j) {
    const int T = j >> 1;
    if ((j & 1) && T + 1 < nkt) wait_younger(min(nkt - 1, T + NST - 1) - (T + 1));  // tile T+1 visible after barrier 2T+1
    __builtin_amdgcn_s_barrier();
    if ((j & 1) && T + NST < nkt) issue(T + NST);  // slot of tile T: both groups hold its fragments
    if ((j & 1) == grp) {
      __builtin_amdgcn_s_setprio(1);
      mfmas();
      __builtin_amdgcn_s_setprio(0);
    } else {
      const int TL = (j + 1 - grp) >> 1;
      if (TL < nkt) {
        load_frags(TL);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  if (d.stamps && tid == 0) {
    d.stamps[blockIdx.x * 4 + 0] = t0c;
    d.stamps[blockIdx.x * 4 + 1] = t0r;
    d.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
    d.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int tm = 0; tm < MTW; ++tm) {
    const int m = m0 + my_row0 + tm * 16 + (lane & 15);
    if (tm >= my_mt || m >= m_end) continue;
    if (SWIGLU) {
#pragma unroll
      for (int tn = 0; tn < NTW; tn += 2) {
        const int n = n0 + wave_n * 64 + tn * 16 + nq;
        if (n >= d.N) continue;
        const f32x4 v = acc[tn][tm], g = acc[tn + 1][tm];
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(unsigned short, (_Float16)(v[r] * (g[r] / (1.f + __expf(-g[r])))));
        const int no = (n0 + wave_n * 64 + tn * 16) / 2 + nq;
        *reinterpret_cast<uint2*>(d.C + (long)m * (d.N / 2) + no) =
            uint2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < NTW; ++tn) {
        const int n = n0 + wave_n * 64 + tn * 16 + nq;
        if (n >= d.N) continue;
        const f32x4 v = acc[tn][tm];
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(unsigned short, (_Float16)v[r]);
        *reinterpret_cast<uint2*>(d.C + (long)m * d.N + n) =
            uint2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
      }
    }
  }
}

__global__ void fill_kernel(op16_t* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    const float v = ((x & 0xffff) / 65536.f - 0.5f) * 0.25f;
    p[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
  }
}

struct Bufs {
  op16_t *A, *C, *zero;
  unsigned long long* stamps;
  std::vector<op16_t*> W;
  int rot = 24;  // rotate over this many weight buffers (each launch uses the next one)
  int pf = 0;
};

template <int WM, int WN, int MT, int NTW, int TBK, int NST, int MODE, int SWIGLU>
float run_cfg(const char* name, Bufs& b, int M, int N, int K, int panel_rows, int iters) {
  constexpr int TBN = WN * NTW * 16;
  const size_t smem = (size_t)NST * (MT * 16 + TBN) * TBK * sizeof(op16_t);
  if (smem > 160 * 1024) {
    printf("%-44s smem %zu too large\n", name, smem);
    return 0;
  }
  auto kern = lab_kernel<WM, WN, MT, NTW, TBK, NST, MODE, SWIGLU>;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  Prob d;
  d.A = b.A;
  d.C = b.C;
  d.M = M;
  d.N = N;
  d.K = K;
  d.panel_rows = panel_rows;
  d.tiles_m = (M + panel_rows - 1) / panel_rows;
  d.tiles_n = (N + TBN - 1) / TBN;
  d.m_fast = 1;
  d.pf_mode = b.pf;
  d.stamps = b.stamps;
  d.wnext_bytes = (long)N * K * 2;
  const int grid = d.tiles_m * d.tiles_n;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) {
    d.W = b.W[w % b.rot];
    d.Wnext = b.W[(w + 1) % b.rot];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, 0, d, b.zero);
  }
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) {
    d.W = b.W[i % b.rot];
    d.Wnext = b.W[(i + 1) % b.rot];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, 0, d, b.zero);
  }
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  CHK(hipGetLastError());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  // main-loop clock and cycles of the last launch (median over workgroups)
  std::vector<unsigned long long> st((size_t)grid * 4);
  CHK(hipMemcpy(st.data(), b.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, ghz;
  for (int g = 0; g < grid; ++g) {
    const double dc = (double)(st[g * 4 + 2] - st[g * 4 + 0]), dr = (double)(st[g * 4 + 3] - st[g * 4 + 1]);
    if (dr > 0) {
      cyc.push_back(dc);
      ghz.push_back(dc / dr * 0.1);
    }
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  const double mcyc = cyc.empty() ? 0 : cyc[cyc.size() / 2], mghz = ghz.empty() ? 0 : ghz[ghz.size() / 2];
  const double tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  const double stage_gb = (double)grid * (MT * 16 + TBN) * K * 2.0 / 1e9;
  printf("%-44s pf %d rot %2d grid %4d wg %4d smem %6zu  %7.2f us  %6.0f TF  staged %.0f MB = %.1f GB/s/CU  loop %.0f cyc @ %.2f GHz\n",
         name, b.pf, b.rot, grid, WM * WN * 64, smem, us, MODE == 1 ? 0.0 : tf, stage_gb * 1e3,
         stage_gb / (us * 1e-6) / grid, mcyc, mghz);
  fflush(stdout);
  return (float)us;
}

// alternate two different kernels (FF-in, QKV) the way the pipeline does: instruction cache, scalar cache and L2
// contents change hands at every launch; optionally a small writer kernel touches A in between (A freshly produced)
static void run_mix(Bufs& b, int M, int iters, int refill) {
  auto k1 = lab_kernel<4, 4, 17, 4, 64, 2, 0, 1>;
  auto k2 = lab_kernel<4, 4, 7, 4, 64, 3, 0, 0>;
  const size_t sm1 = (size_t)2 * (17 * 16 + 256) * 64 * 2, sm2 = (size_t)3 * (7 * 16 + 256) * 64 * 2;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  Prob d1, d2;
  d1.A = b.A; d1.C = b.C; d1.M = M; d1.N = 8192; d1.K = 1024; d1.panel_rows = 264; d1.tiles_m = 8; d1.tiles_n = 32;
  d1.m_fast = 1; d1.pf_mode = 0; d1.stamps = b.stamps; d1.wnext_bytes = 0;
  d2 = d1; d2.N = 3072; d2.panel_rows = 104; d2.tiles_m = (M + 103) / 104; d2.tiles_n = 12;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {  // 0: k1 only, 1: k2 only, 2: alternating
    CHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) {
      d1.W = b.W[i % b.rot]; d1.Wnext = d1.W;
      d2.W = b.W[(i + 7) % b.rot]; d2.Wnext = d2.W;
      if (refill == 1) fill_kernel<<<256, 256>>>(b.A, (long)M * 1024, 1 + i);
      if (refill == 2) fill_kernel<<<2048, 256>>>(b.C, (long)M * 8192, 1 + i);  // 35 MB streamed: an HBM-bound neighbour
      if (mode != 1) hipLaunchKernelGGL(k1, dim3(256), dim3(1024), sm1, 0, d1, b.zero);
      if (refill == 1 && mode == 2) fill_kernel<<<256, 256>>>(b.A, (long)M * 1024, 77 + i);
      if (refill == 2 && mode == 2) fill_kernel<<<2048, 256>>>(b.C, (long)M * 8192, 77 + i);
      if (mode != 0) hipLaunchKernelGGL(k2, dim3(252), dim3(1024), sm2, 0, d2, b.zero);
    }
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double mghz = 0;
    {  // shader clock inside the k loop of the last launch (median over workgroups)
      const int grid = mode == 1 ? 252 : 252;
      std::vector<unsigned long long> st((size_t)grid * 4);
      CHK(hipMemcpy(st.data(), b.stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> ghz;
      for (int g = 0; g < grid; ++g) {
        const double dc = (double)(st[g * 4 + 2] - st[g * 4 + 0]), dr = (double)(st[g * 4 + 3] - st[g * 4 + 1]);
        if (dr > 0) ghz.push_back(dc / dr * 0.1);
      }
      std::sort(ghz.begin(), ghz.end());
      if (!ghz.empty()) mghz = ghz[ghz.size() / 2];
    }
    printf("mix refill %d mode %d (%s): %.2f us per iteration, last k loop @ %.2f GHz\n", refill, mode, mode == 0 ? "ffin only" : mode == 1 ? "qkv only" : "ffin+qkv alternating", 1e3 * ms / iters, mghz);
  }
  if (refill) {
    CHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) {
      if (refill == 1) fill_kernel<<<256, 256>>>(b.A, (long)M * 1024, 1 + i);
      else fill_kernel<<<2048, 256>>>(b.C, (long)M * 8192, 1 + i);
    }
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("mix: writer kernel alone %.2f us\n", 1e3 * ms / iters);
  }
}

static std::vector<unsigned short> g_ref;  // output of the last baseline run (for checking new kernels)
static void snapshot(const Bufs& b, long n, std::vector<unsigned short>& out) {
  out.resize((size_t)n);
  CHK(hipMemcpy(out.data(), b.C, (size_t)n * 2, hipMemcpyDeviceToHost));
}
static float h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

template <int MT, int NST, int SWIGLU>
float run_pp(const char* name, Bufs& b, int M, int N, int K, int panel_rows, int iters, bool check) {
  const size_t smem = (size_t)NST * (MT * 16 + 256) * 32 * sizeof(op16_t);
  if (smem > 160 * 1024) {
    printf("%-44s smem %zu too large\n", name, smem);
    return 0;
  }
  auto kern = pp_kernel<MT, NST, SWIGLU>;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  Prob d;
  d.A = b.A;
  d.C = b.C;
  d.M = M;
  d.N = N;
  d.K = K;
  d.panel_rows = panel_rows;
  d.tiles_m = (M + panel_rows - 1) / panel_rows;
  d.tiles_n = (N + 255) / 256;
  d.m_fast = 1;
  d.pf_mode = 0;
  d.wnext_bytes = 0;
  d.Wnext = nullptr;
  d.stamps = b.stamps;
  const int grid = d.tiles_m * d.tiles_n;
  if (check) {  // same weights as the baseline's last launch: buffer 0
    CHK(hipMemset(b.C, 0, (size_t)M * N * 2));
    d.W = b.W[0];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, d, b.zero);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned short> got;
    const long n = (long)M * (SWIGLU ? N / 2 : N);
    snapshot(b, n, got);
    double maxd = 0, maxr = 0;
    for (long i = 0; i < n && i < (long)g_ref.size(); ++i) {
      maxd = std::max(maxd, (double)fabsf(h2f(got[i]) - h2f(g_ref[i])));
      maxr = std::max(maxr, (double)fabsf(h2f(g_ref[i])));
    }
    printf("%-44s check vs baseline: max |diff| %.4g (max |ref| %.4g) %s\n", name, maxd, maxr, maxd <= 2e-3 * maxr + 1e-6 ? "OK" : "MISMATCH");
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) {
    d.W = b.W[w % b.rot];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, d, b.zero);
  }
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) {
    d.W = b.W[i % b.rot];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, d, b.zero);
  }
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  CHK(hipGetLastError());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  std::vector<unsigned long long> st((size_t)grid * 4);
  CHK(hipMemcpy(st.data(), b.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, ghz;
  for (int g = 0; g < grid; ++g) {
    const double dc = (double)(st[g * 4 + 2] - st[g * 4 + 0]), dr = (double)(st[g * 4 + 3] - st[g * 4 + 1]);
    if (dr > 0) {
      cyc.push_back(dc);
      ghz.push_back(dc / dr * 0.1);
    }
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  printf("%-44s pp rot %2d grid %4d smem %6zu  %7.2f us  %6.0f TF  loop %.0f cyc @ %.2f GHz\n", name, b.rot, grid, smem, us,
         2.0 * M * N * K / (us * 1e-6) / 1e12, cyc.empty() ? 0 : cyc[cyc.size() / 2], ghz.empty() ? 0 : ghz[ghz.size() / 2]);
  fflush(stdout);
  return (float)us;
}

int main(int argc, char** argv) {
  const int M = 2112;
  const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 48;
  const int NW = 64;  // weight buffers (16.8 MB each: 1 GB)
  Bufs b;
  const long maxA = (long)M * 4096, maxW = 8192L * 1024, maxC = (long)M * 8192;
  CHK(hipMalloc((void**)&b.A, maxA * 2));
  CHK(hipMalloc((void**)&b.C, maxC * 2));
  CHK(hipMalloc((void**)&b.zero, 4096));
  CHK(hipMalloc((void**)&b.stamps, 4096 * 4 * 8));
  CHK(hipMemset(b.stamps, 0, 4096 * 4 * 8));
  CHK(hipMemset(b.zero, 0, 4096));
  fill_kernel<<<1024, 256>>>(b.A, maxA, 1);
  for (int i = 0; i < NW; ++i) {
    op16_t* w;
    CHK(hipMalloc((void**)&w, maxW * 2));
    fill_kernel<<<1024, 256>>>(w, maxW, 100 + i);
    b.W.push_back(w);
  }
  CHK(hipDeviceSynchronize());
  const char* only = argc > 1 ? argv[1] : "";
  auto want = [&](const char* tag) { return only[0] == 0 || strstr(tag, only) != nullptr; };

#define RUN(tag, WM, WN, MT, NTW, BK, NST, MODE, SW, M_, N_, K_, PR) \
  if (want(tag)) run_cfg<WM, WN, MT, NTW, BK, NST, MODE, SW>(tag, b, M_, N_, K_, PR, iters);

  if (want("mix")) {
    b.rot = 64;
    run_mix(b, M, iters, 0);
    run_mix(b, M, iters, 1);
    run_mix(b, M, iters, 2);
  }
  b.pf = 0;
#define PP(tag, MT, NST, SW, M_, N_, K_, PR, CHECK) \
  if (want(tag)) run_pp<MT, NST, SW>(tag, b, M_, N_, K_, PR, iters, CHECK);
  // ---- FF-in: baseline once with weight buffer 0 last (reference output), then the ping-pong kernel
  b.rot = 1;
  RUN("ffin 16w 4x4 mt17 bk64 nst2 full", 4, 4, 17, 4, 64, 2, 0, 1, M, 8192, 1024, 264)
  snapshot(b, (long)M * 4096, g_ref);
  PP("ffin pp mt17 nst4", 17, 4, 1, M, 8192, 1024, 264, true)
  b.rot = 64;
  RUN("ffin 16w 4x4 mt17 bk64 nst2 full", 4, 4, 17, 4, 64, 2, 0, 1, M, 8192, 1024, 264)
  PP("ffin pp mt17 nst4", 17, 4, 1, M, 8192, 1024, 264, false)
  PP("ffin pp mt16 nst4 (256-row tiles, 9 panels)", 16, 4, 1, M, 8192, 1024, 256, false)
  // ---- QKV
  b.rot = 1;
  RUN("qkv 16w 4x4 mt7 bk64 nst3 full", 4, 4, 7, 4, 64, 3, 0, 0, M, 3072, 1024, 104)
  snapshot(b, (long)M * 3072, g_ref);
  PP("qkv pp mt7 nst4", 7, 4, 0, M, 3072, 1024, 104, true)
  b.rot = 64;
  RUN("qkv 16w 4x4 mt7 bk64 nst3 full", 4, 4, 7, 4, 64, 3, 0, 0, M, 3072, 1024, 104)
  PP("qkv pp mt7 nst4", 7, 4, 0, M, 3072, 1024, 104, false)
  PP("qkv pp mt9 nst4 (132-row panels, 192 wgs)", 9, 4, 0, M, 3072, 1024, 132, false)
  return 0;
}
