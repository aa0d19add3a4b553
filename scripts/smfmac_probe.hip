// Operand layout of v_smfmac_i32_32x32x64_i8 on gfx950 (2:4 structured-sparse int8 MFMA), found by experiment: groundwork for
// running the missing-mask product of the exact int8 U^T x at twice the dense rate.  Not part of the
// library.   hipcc --offload-arch=gfx950 -O2 scripts/smfmac_probe.hip -o /tmp/smfmac_probe && /tmp/smfmac_probe
//
// The instruction computes C (32 x 32, i32) += A (32 x 64 logical, 2 of every 4 consecutive k kept) * B (64 x 32).  Per lane:
// A = 16 kept bytes, B = 32 bytes, idx = 32 bits.  The probe assumes lane l holds row / column l % 32 and the k range
// 32 (l / 32) .. + 31 of both operands (B byte q of the lane = k 32 (l/32) + q), puts B[k][j] = k + 1 for every column, a single
// 1 into kept slot p of every lane and prints which k each slot selects for a few index patterns: the value of C[row][col]
// is then (selected k) + 1, summed over the two lane halves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k64(const v4i *a, const v8i *b, const int *idx, v16i *c) {
  const int l = threadIdx.x;
  v16i acc = {0};
  acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[l], b[l], acc, idx[l], 0, 0);
  c[l] = acc;
}

int main() {
  signed char hA[64][16], hB[64][32];
  int hI[64];
  int hC[64][16];
  void *dA, *dB, *dI, *dC;
  hipMalloc(&dA, sizeof hA);
  hipMalloc(&dB, sizeof hB);
  hipMalloc(&dI, sizeof hI);
  hipMalloc(&dC, sizeof hC);
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 32; ++q) hB[l][q] = (signed char)(32 * (l / 32) + q + 1); // B[k][j] = k + 1, any column
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  const unsigned pats[] = {0x00000000u, 0x44444444u, 0xEEEEEEEEu, 0x88888888u, 0xCCCCCCCCu, 0x4E4E4E4Eu, 0x12345678u};
  for (unsigned pat : pats) {
    printf("idx = 0x%08x\n", pat);
    for (int half = 0; half < 2; ++half) {
      for (int p = 0; p < 16; ++p) {
        memset(hA, 0, sizeof hA);
        for (int l = 32 * half; l < 32 * half + 32; ++l) hA[l][p] = 1; // only one lane half is non-zero
        for (int l = 0; l < 64; ++l) hI[l] = (int)pat;
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
        hipMemcpy(dI, hI, sizeof hI, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k64, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v8i *)dB, (const int *)dI, (v16i *)dC);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // every C entry should be the same number (all rows identical, all columns identical): print lane 0 reg 0, and flag if not
        bool uniform = true;
        for (int l = 0; l < 64; ++l)
          for (int r = 0; r < 16; ++r) uniform = uniform && hC[l][r] == hC[0][0];
        printf("  lanes %2d..%2d slot %2d -> k = %3d%s\n", 32 * half, 32 * half + 31, p, hC[0][0] - 1, uniform ? "" : "  (C not uniform)");
      }
    }
  }
  // which lanes / registers hold which C[row][col]: A selects row r only (slot 0 = 1 in lanes r and r + 32 -> one k per half),
  // B = 1 in column c only
  printf("accumulator layout (row, col) -> (lane, reg):\n");
  for (int rc = 0; rc < 6; ++rc) {
    const int row = (rc * 7 + 3) % 32, col = (rc * 11 + 5) % 32;
    memset(hA, 0, sizeof hA);
    memset(hB, 0, sizeof hB);
    hA[row][0] = 1;
    for (int q = 0; q < 32; ++q) hB[col][q] = 1;
    for (int l = 0; l < 64; ++l) hI[l] = 0x44444444;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dI, hI, sizeof hI, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k64, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v8i *)dB, (const int *)dI, (v16i *)dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r)
        if (hC[l][r]) printf("  (%2d, %2d) -> lane %2d reg %2d value %d\n", row, col, l, r, hC[l][r]);
  }
  return 0;
}
