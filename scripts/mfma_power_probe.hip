// Sustained rate of the matrix instructions the exact U^T x could run on, with operands held in registers (no memory traffic at
// all) and DIFFERENT VALUE PATTERNS in them: under the power limit the achieved rate depends on the operands' switching activity
// (profiles/r04_i8_operand_value_power.txt), so "which instruction and which digit encoding" is a measurement.  Not part of the
// library.   hipcc --offload-arch=gfx950 -O2 scripts/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
//
// Instructions: v_mfma_i32_32x32x32_i8 (the genotype product today), v_mfma_i32_16x16x64_i8, v_smfmac_i32_32x32x64_i8 (the mask
// product today), v_mfma_f32_32x32x64_f8f6f4 with e4m3 operands (twice the int8 rate; integers 0..15 are exact in e4m3 and fp32
// accumulation is exact below 2^24).  Left operand: genotype-like values (0, 1, 2 with probabilities 0.5, 0.35, 0.15).  Right
// operand ("digits"): mode 0 uniform in [-128, 127]; 1 all zero; 2 uniform in [0, 15]; 3 uniform in [-8, 7]; 4 uniform in [0, 127].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "smi_sampler.hpp" // SMI=1: socket power / clock / power-limit residency per run (use iters >= 400000)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ inline int geno_byte(unsigned h) { const unsigned u = h % 100; return u < 50 ? 0 : (u < 85 ? 1 : 2); }
__device__ inline int digit_byte(unsigned h, int mode) {
  const int hv = (int)(h & 255);
  if (mode == 1) return 0;
  if (mode == 2) return hv & 15;
  if (mode == 3) return (hv & 15) - 8;
  if (mode == 4) return hv & 127;
  return hv - 128;
}
// e4m3 (bias 7) encodings of the integers 0..15
__device__ inline int e4m3_of(int v) {
  const int tab[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};
  return tab[v & 15];
}
template <int NW> __device__ inline void fill(int (&w)[NW], unsigned seed, int mode, bool geno, bool fp8) {
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    int word = 0;
    for (int b = 0; b < 4; ++b) {
      const unsigned h = hash32(seed * 977u + (unsigned)(i * 4 + b) * 40503u + 12345u);
      int v = geno ? geno_byte(h) : digit_byte(h, mode);
      if (fp8) v = e4m3_of(geno ? v : (mode == 1 ? 0 : (int)(h & 15)));
      word |= (v & 255) << (8 * b);
    }
    w[i] = word;
  }
}

// KIND 0: i8 32x32x32; 1: i8 16x16x64; 2: sparse i8 32x32x64; 3: fp8 32x32x64
template <int KIND> __global__ __launch_bounds__(512) void probe(int iters, int mode, int *sink, int a_zero) {
  const unsigned seed = blockIdx.x * 512u + threadIdx.x;
  int a4[4], b4[4], a8[8], b8[8];
  fill<4>(a4, seed, mode, true, false);
  fill<4>(b4, seed + 7919u, mode, false, false);
  fill<8>(a8, seed, mode, true, KIND == 3);
  fill<8>(b8, seed + 7919u, mode, false, KIND == 3);
  v4i A4 = {a4[0], a4[1], a4[2], a4[3]}, B4 = {b4[0], b4[1], b4[2], b4[3]};
  v8i A8 = {a8[0], a8[1], a8[2], a8[3], a8[4], a8[5], a8[6], a8[7]}, B8 = {b8[0], b8[1], b8[2], b8[3], b8[4], b8[5], b8[6], b8[7]};
  if (a_zero) { // A_ZERO=1: the left operand of the real mask product -- kept values almost all zero (one lane in 16 holds a single 1)
    const int one = (threadIdx.x & 15) == 3 ? 1 : 0;
    A4 = (v4i){one, 0, 0, 0};
    A8 = (v8i){one, 0, 0, 0, 0, 0, 0, 0};
  }
  const int idx = 0x44444444;
  v16i c[8];
  v16f f[8];
  v4i s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) { c[q][r] = 0; f[q][r] = 0.0f; }
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (KIND == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c[q]) : "v"(A4), "v"(B4));
      if (KIND == 1) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(s[q]) : "v"(A4), "v"(B4));
      if (KIND == 2) asm volatile("v_smfmac_i32_32x32x64_i8 %0, %1, %2, %3" : "+v"(c[q]) : "v"(A4), "v"(B8), "v"(idx));
      if (KIND == 3) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(f[q]) : "v"(A8), "v"(B8));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  int acc = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += c[q][r] + (int)f[q][r] + s[q][r & 3];
  if (acc == 0x7fffffff) sink[0] = acc;
}

// exactness of the fp8 form on small integers: A = 2 everywhere, B = 3 everywhere -> every C entry 64 * 6 = 384
__global__ void fp8_check(float *out) {
  v8i A, B;
  for (int i = 0; i < 8; ++i) { A[i] = 0x40404040; B[i] = 0x44444444; }
  v16f c;
  for (int r = 0; r < 16; ++r) c[r] = 0.0f;
  asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(c) : "v"(A), "v"(B));
  out[threadIdx.x] = c[0] + c[15];
}

template <int KIND> static double run(int iters, int mode, int *sink, double ops_per_instr) {
  const int grid = 2048; // 8 wavefronts per workgroup
  const int a_zero = getenv("A_ZERO") ? atoi(getenv("A_ZERO")) : 0;
  hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(512), 0, 0, iters / 8, mode, sink, a_zero);
  hipDeviceSynchronize();
  static SmiSampler smi;
  static int smi_state = -1;
  if (smi_state < 0) smi_state = (getenv("SMI") && atoi(getenv("SMI")) && smi.open()) ? 1 : 0;
  if (smi_state) smi.start();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(512), 0, 0, iters, mode, sink, a_zero);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  if (smi_state) {
    char tag[64];
    snprintf(tag, sizeof tag, "kind %d digits-mode %d a_zero %d", KIND, mode, a_zero);
    smi.stop(tag);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)grid * 8 * iters * 8 * ops_per_instr / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  int *sink;
  hipMalloc(&sink, 64);
  float *chk, hchk[64];
  hipMalloc(&chk, 64 * sizeof(float));
  hipLaunchKernelGGL(fp8_check, dim3(1), dim3(64), 0, 0, chk);
  hipMemcpy(hchk, chk, sizeof hchk, hipMemcpyDeviceToHost);
  printf("fp8 (e4m3) check: 2 x 3 over K = 64 -> %.1f + %.1f (expected 384 + 384)\n", hchk[0] / 2, hchk[0] / 2);
  const char *names[5] = {"uniform [-128, 127]", "all zero", "uniform [0, 15]", "uniform [-8, 7]", "uniform [0, 127]"};
  printf("sustained TOP/s (2 x M x N x K per instruction; the sparse form counted at its logical K = 64), left operand 0/1/2:\n");
  for (int mode = 0; mode < 5; ++mode) {
    const double r0 = run<0>(iters, mode, sink, 2.0 * 32 * 32 * 32);
    const double r1 = run<1>(iters, mode, sink, 2.0 * 16 * 16 * 64);
    const double r2 = run<2>(iters, mode, sink, 2.0 * 32 * 32 * 64);
    printf("  digits %-20s  i8 32x32x32 %7.1f   i8 16x16x64 %7.1f   sparse i8 32x32x64 %7.1f", names[mode], r0, r1, r2);
    if (mode == 0 || mode == 1 || mode == 2) {
      const double r3 = run<3>(iters, mode, sink, 2.0 * 32 * 32 * 64);
      printf("   fp8 32x32x64 (digits %s) %7.1f", mode == 1 ? "zero" : "0..15", r3);
    }
    printf("\n");
  }
  return 0;
}
