// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate of the chip (register-only, no memory) and
// the shader clock it runs at (s_memtime ticks / wall time).  Inline asm pins the accumulators in
// VGPRs (hipcc's builtin version shuffles them through AGPRs every iteration and measures the copies).
// The datasheet 78.6 TFLOP/s assumes 2.4 GHz and 64 cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters, double a0, double astep, double b0,
                                        double bstep) {
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  double a = a0 + threadIdx.x * astep, b = b0 + threadIdx.x * bstep;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    MF(c0); MF(c1); MF(c2); MF(c3); MF(c4); MF(c5); MF(c6); MF(c7);
    MF(c0); MF(c1); MF(c2); MF(c3); MF(c4); MF(c5); MF(c6); MF(c7);
  }
  long long t1 = clock64();
  f64x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
void run(const char *tag, int blocks_per_cu, int iters, double a0, double astep, double b0, double bstep) {
  int blocks = 256 * blocks_per_cu;
  double *out;
  long long *cyc, hc[4];
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 10, a0, astep, b0, bstep);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, a0, astep, b0, bstep);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  const double n_mfma = 16.0 * iters;
  double flops = 2.0 * 16 * 16 * 4 * n_mfma * 4.0 * blocks;
  printf("%-24s waves/SIMD=%d: %7.1f ms %6.2f TFLOP/s | %.1f s_memtime ticks per MFMA per wave, %.0f ticks/us\n", tag,
         blocks_per_cu, ms, flops / ms / 1e9, (double)hc[0] / n_mfma, (double)hc[0] / (ms * 1e3));
  hipFree(out); hipFree(cyc);
}
int main() {
  run("zeros", 1, 200000, 0.0, 0.0, 0.0, 0.0);
  run("zeros", 2, 200000, 0.0, 0.0, 0.0, 0.0);
  run("genotype-like a, rnd b", 2, 200000, 1.0, 0.0, 0.3712894651, 1.23456789e-3);
  run("random-ish a and b", 1, 200000, 0.7312345678, 3.3333331e-4, 0.3712894651, 1.23456789e-3);
  run("random-ish a and b", 2, 200000, 0.7312345678, 3.3333331e-4, 0.3712894651, 1.23456789e-3);
  run("random-ish a and b", 4, 100000, 0.7312345678, 3.3333331e-4, 0.3712894651, 1.23456789e-3);
  return 0;
}
