// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate of the chip (register-only, no memory).
// Establishes the empirical ceiling the GEMM's roofline fraction should also be read against
// (the datasheet 78.6 TFLOP/s assumes 2.4 GHz; the sustained clock under fp64 MFMA load is lower).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  int ncu = 256;
  int blocks = ncu * blocks_per_cu;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;
  printf("nacc=%d blocks/CU=%d: %.1f ms, %.2f TFLOP/s, %.1f cyc/MFMA/SIMD @2.4GHz-equivalent\n", NACC, blocks_per_cu, ms,
         flops / ms / 1e9, 2.4e9 * ms * 1e-3 / ((double)NACC * iters * blocks_per_cu));
  hipFree(out);
}
int main() {
  run<4>(1, 200000);
  run<16>(1, 50000);
  run<16>(2, 50000);
  run<8>(4, 50000);
  run<16>(2, 400000);  // ~ 1 s sustained
  return 0;
}
