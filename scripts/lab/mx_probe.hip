// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales) on gfx950:
// checks the operand lane map, the per-lane scale association and the C/D map with exact small-integer data.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/lab/mx_probe scripts/lab/mx_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB,
                      float* D) {
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  i32x8 a = *reinterpret_cast<const i32x8*>(A + r * 128 + q * 32);
  i32x8 b = *reinterpret_cast<const i32x8*>(B + r * 128 + q * 32);
  const int sa = SA[r * 4 + q], sb = SB[r * 4 + q];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int t = 0; t < 4; ++t) D[(4 * q + t) * 16 + r] = c[t];   // D[i = 4q+t][j = r]
}

static unsigned char enc(int v) {   // small integers as OCP e4m3
  static const unsigned char tab[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return v < 0 ? (unsigned char)(0x80 | tab[-v]) : tab[v];
}

int main() {
  unsigned char hA[16 * 128], hB[16 * 128], hSA[64], hSB[64];
  int iA[16][128], iB[16][128];
  srand(3);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 128; ++k) {
      iA[i][k] = rand() % 9 - 4;
      iB[i][k] = rand() % 7 - 3;
      hA[i * 128 + k] = enc(iA[i][k]);
      hB[i * 128 + k] = enc(iB[i][k]);
    }
  unsigned char *dA, *dB, *dSA, *dSB;
  float* dD;
  hipMalloc((void**)&dA, sizeof hA);
  hipMalloc((void**)&dB, sizeof hB);
  hipMalloc((void**)&dSA, 64);
  hipMalloc((void**)&dSB, 64);
  hipMalloc((void**)&dD, 256 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 64; ++i) {
      hSA[i] = pass ? 127 + (i * 7) % 5 - 2 : 127;
      hSB[i] = pass ? 127 + (i * 3) % 4 - 1 : 127;
    }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice);
    hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dSA, dSB, dD);
    float hD[256];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int qb = 0; qb < 4; ++qb) {
          double s = 0;
          for (int k = 0; k < 32; ++k) s += iA[i][qb * 32 + k] * iB[j][qb * 32 + k];
          ref += s * ldexp(1.0, hSA[i * 4 + qb] - 127) * ldexp(1.0, hSB[j * 4 + qb] - 127);
        }
        if (fabs(ref - hD[i * 16 + j]) > 1e-3) {
          if (bad < 5) printf("  mismatch D[%d][%d] = %g, want %g\n", i, j, hD[i * 16 + j], ref);
          ++bad;
        }
      }
    printf("pass %d (%s scales): %d / 256 mismatches\n", pass, pass ? "per-block" : "unit", bad);
  }
  // which scale lane covers which 16-byte half of which data lane?  A = B = 1 except one half-lane of A = 2
  for (int q0 = 0; q0 < 4; ++q0)
    for (int half = 0; half < 2; ++half) {
      for (int i = 0; i < 16 * 128; ++i) hA[i] = hB[i] = enc(1);
      for (int j = 0; j < 16; ++j) hA[0 * 128 + q0 * 32 + half * 16 + j] = enc(2);   // row 0, data lane (0, q0)
      hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
      hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
      printf("data lane q=%d half %d = 2:", q0, half);
      for (int q1 = 0; q1 < 4; ++q1) {
        for (int i = 0; i < 64; ++i) hSA[i] = hSB[i] = 127;
        hSA[0 * 4 + q1] = 130;
        hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice);
        hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dSA, dSB, dD);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        printf("  scale lane q=%d -> D[0][0]=%g", q1, hD[0]);   // 144 + 224 + 7*16*(covered ? 1 : 0)
      }
      printf("\n");
    }
  return 0;
}
