// Operand / accumulator layout of v_mfma_i32_16x16x64_i8 and the lane exchange of v_permlane16_swap_b32 on gfx950, found by
// experiment: groundwork for moving the genotype product of the records kernel (i8gemm_sparse2.hip.h) to the 16x16x64 form,
// which keeps 4.72 POP/s on full-range digit values where the 32x32x32 form drops to 3.70 (profiles/r04_mfma_power_probe.txt).
// Not part of the library.   hipcc --offload-arch=gfx950 -O2 scripts/mfma16_layout_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k16(const v4i *a, const v4i *b, v4i *c) {
  const int l = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(acc) : "v"(a[l]), "v"(b[l]));
  c[l] = acc;
}
// x, y: one dword per lane -> after v_permlane16_swap_b32 x, y
__global__ void kswap(int *x, int *y) {
  const int l = threadIdx.x;
  int vx = x[l], vy = y[l];
  asm volatile("v_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(vx), "+v"(vy));
  x[l] = vx;
  y[l] = vy;
}

int main() {
  signed char hA[64][16], hB[64][16];
  int hC[64][4];
  void *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA);
  hipMalloc(&dB, sizeof hB);
  hipMalloc(&dC, sizeof hC);
  // (1) which (row, k) a byte of A is: A = one 1 at (lane la, byte ba); B[k][j] = k + 1 for every column (assuming B's lane l holds
  //     column l % 16 and k = 16 (l / 16) + byte -- checked in (2)); C[row][*] = k + 1
  printf("A operand: (lane, byte) -> (row, k), assuming B lane l = column l %% 16, k = 16 (l / 16) + byte\n");
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 16; ++q) hB[l][q] = (signed char)(16 * (l / 16) + q + 1);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  const int la_list[6] = {0, 5, 16, 21, 37, 63}, ba_list[3] = {0, 7, 15};
  for (int la : la_list)
    for (int ba : ba_list) {
      memset(hA, 0, sizeof hA);
      hA[la][ba] = 1;
      hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v4i *)dB, (v4i *)dC);
      hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
      int cnt = 0, val = 0, lane0 = -1, reg0 = -1;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
          if (hC[l][r]) { if (!cnt) { val = hC[l][r]; lane0 = l; reg0 = r; } ++cnt; }
      printf("  A(lane %2d, byte %2d): %2d non-zero C entries, value %3d (k = %3d), first at lane %2d reg %d\n", la, ba, cnt, val, val - 1,
             lane0, reg0);
    }
  // (2) accumulator layout: A row i = all ones over K (assuming lane l = row l % 16), B column j = all ones -> C[i][j] = 64
  printf("accumulator: (row, col) -> (lane, reg), assuming A lane l = row l %% 16 and B lane l = column l %% 16\n");
  for (int t = 0; t < 6; ++t) {
    const int row = (t * 5 + 2) % 16, col = (t * 7 + 3) % 16;
    memset(hA, 0, sizeof hA);
    memset(hB, 0, sizeof hB);
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < 16; ++q) {
        if (l % 16 == row) hA[l][q] = 1;
        if (l % 16 == col) hB[l][q] = 1;
      }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v4i *)dB, (v4i *)dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r)
        if (hC[l][r]) printf("  (%2d, %2d) -> lane %2d reg %d value %d\n", row, col, l, r, hC[l][r]);
  }
  // (3) k pairing: A = 1 at (lane la, byte ba) only, B = 1 at (lane lb, byte bb) only: non-zero iff they are the same k (and then
  //     C[row la %16][col lb % 16] = 1): which (lb, bb) pair with A's (16, 3)?
  printf("k pairing: A(lane 16, byte 3) against B(lane lb, byte 3):");
  for (int lb = 0; lb < 64; lb += 16) {
    memset(hA, 0, sizeof hA);
    memset(hB, 0, sizeof hB);
    hA[16][3] = 1;
    hB[lb][3] = 1;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, (const v4i *)dA, (const v4i *)dB, (v4i *)dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int s = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) s += hC[l][r];
    printf("  lb %2d -> %d", lb, s);
  }
  printf("\n");
  // (4) v_permlane16_swap_b32 x, y with x[l] = l, y[l] = 100 + l
  int hx[64], hy[64], *dx, *dy;
  for (int l = 0; l < 64; ++l) { hx[l] = l; hy[l] = 100 + l; }
  hipMalloc(&dx, sizeof hx);
  hipMalloc(&dy, sizeof hy);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
  hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kswap, dim3(1), dim3(64), 0, 0, dx, dy);
  hipMemcpy(hx, dx, sizeof hx, hipMemcpyDeviceToHost);
  hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost);
  printf("v_permlane16_swap_b32 x, y (x = lane, y = 100 + lane before):\n  x:");
  for (int l = 0; l < 64; ++l) printf(" %3d", hx[l]);
  printf("\n  y:");
  for (int l = 0; l < 64; ++l) printf(" %3d", hy[l]);
  printf("\n");
  return 0;
}
