// Operand layout of v_smfmac_i32_16x16x128_i8 on gfx950 (2:4 structured-sparse int8, 16 x 16 output, logical K = 128), found by
// experiment: groundwork for moving the mask product of the records kernel next to a genotype product on v_mfma_i32_16x16x64_i8
// (DESIGN.md section 10).  Not part of the library.   hipcc --offload-arch=gfx950 -O2 scripts/smfmac16_layout_probe.hip -o /tmp/p
//
// A: 16 kept bytes per lane (row = lane % 16 assumed), B: 32 bytes per lane (column = lane % 16 assumed), idx: 32 bits per lane.
// Every B byte carries an ID: lane quarter qb = lane / 16, byte b: ID = 32 qb + b + 1 for qb < 3, -(b + 1) for qb = 3.  A holds a
// single 1 in kept slot p of lane (row 0, quarter qa); C[0][*] is then the ID of the B byte that slot multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void ks(const v4i *a, const v8i *b, const int *idx, v4i *c) {
  const int l = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  asm volatile("v_smfmac_i32_16x16x128_i8 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 7" : "+v"(acc) : "v"(a[l]), "v"(b[l]), "v"(idx[l]));
  c[l] = acc;
}

int main() {
  signed char hA[64][16], hB[64][32];
  int hI[64], hC[64][4];
  void *dA, *dB, *dI, *dC;
  hipMalloc(&dA, sizeof hA);
  hipMalloc(&dB, sizeof hB);
  hipMalloc(&dI, sizeof hI);
  hipMalloc(&dC, sizeof hC);
  for (int l = 0; l < 64; ++l)
    for (int b = 0; b < 32; ++b) hB[l][b] = (signed char)((l / 16) < 3 ? 32 * (l / 16) + b + 1 : -(b + 1));
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  const unsigned pats[3] = {0x44444444u, 0xEEEEEEEEu, 0x88888888u}; // kept positions (0, 1), (2, 3), (0, 2) of every group of four
  for (unsigned pat : pats) {
    printf("idx = 0x%08x: A(quarter qa, kept slot p) -> B byte ID (32 qb + b + 1; negative: qb = 3, -(b + 1))\n", pat);
    for (int qa = 0; qa < 4; ++qa) {
      printf("  qa %d:", qa);
      for (int p = 0; p < 16; ++p) {
        memset(hA, 0, sizeof hA);
        hA[16 * qa][p] = 1;
        for (int l = 0; l < 64; ++l) hI[l] = (int)pat;
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
        hipMemcpy(dI, hI, sizeof hI, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(ks, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v8i *)dB, (const int *)dI, (v4i *)dC);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // row 0 of C: which (lane, reg) holds it is part of the question: print the first non-zero entry and how many there are
        int cnt = 0, val = 0;
        for (int l = 0; l < 64; ++l)
          for (int r = 0; r < 4; ++r)
            if (hC[l][r]) { if (!cnt) val = hC[l][r]; ++cnt; }
        printf(" %4d%s", val, cnt == 16 ? "" : "*");
      }
      printf("\n");
    }
  }
  // accumulator layout: A row i all kept slots 1 (lanes with l % 16 == i), B column j all ones
  printf("accumulator (row, col) -> (lane, reg):\n");
  for (int t = 0; t < 4; ++t) {
    const int row = (t * 5 + 2) % 16, col = (t * 7 + 3) % 16;
    memset(hA, 0, sizeof hA);
    memset(hB, 0, sizeof hB);
    for (int l = 0; l < 64; ++l) {
      if (l % 16 == row) memset(hA[l], 1, 16);
      if (l % 16 == col) memset(hB[l], 1, 32);
      hI[l] = 0x44444444;
    }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dI, hI, sizeof hI, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ks, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v8i *)dB, (const int *)dI, (v4i *)dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r)
        if (hC[l][r]) printf("  (%2d, %2d) -> lane %2d reg %d value %d\n", row, col, l, r, hC[l][r]);
  }
  return 0;
}
