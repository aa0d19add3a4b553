// Micro-benchmark 2: the GEMM's register pattern -- 4x4 accumulator blocks, 4 A and 4 B operand registers,
// every MFMA with a different (a_i, b_j, acc_ij) triple -- at 1, 2 and 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MF(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
template <int TN>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  f64x4 c[4][TN];
  double a[4], b[TN];
  for (int i = 0; i < 4; ++i) {
    a[i] = a0 + threadIdx.x * 1e-3 + i;
    for (int j = 0; j < TN; ++j) c[i][j] = f64x4{0, 0, 0, 0};
  }
  for (int j = 0; j < TN; ++j) b[j] = b0 + threadIdx.x * 1e-4 + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) MF(c[i][j], a[i], b[j]);
  }
  f64x4 s = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < TN; ++j) s += c[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int TN>
void run(int blocks_per_cu, int iters) {
  int blocks = 256 * blocks_per_cu;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<TN>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<TN>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 16 * 16 * 4 * 4.0 * TN * iters * 4.0 * blocks;
  printf("4x%d accumulators, waves/SIMD=%d: %7.1f ms %6.2f TFLOP/s\n", TN, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(1, 100000);
  run<4>(2, 100000);
  run<2>(1, 200000);
  run<2>(2, 200000);
  run<2>(4, 100000);
  return 0;
}
