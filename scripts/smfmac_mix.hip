// The int8 U^T x inner loop as an instruction mix, register operands only (no LDS, no global traffic): per 64 bytes of K a
// wavefront issues 8 dense MFMAs for the genotype product and either 8 dense MFMAs (today) or 4 sparse ones (2:4 compressed
// mask) for the missing-mask product.  512 threads per CU on every CU, like i8gemm_packed_kernel_t.  Upper bound on what the
// sparse mask operand can buy under the chip's power limit.
//   hipcc --offload-arch=gfx950 -O2 scripts/smfmac_mix.hip -o /tmp/smfmac_mix && /tmp/smfmac_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <bool SPARSE>
__global__ __launch_bounds__(512) void mix_kernel(int iters, const int *seed, int *out) {
  const int l = threadIdx.x;
  // operands with realistic bit activity: genotypes 0/1/2, mask mostly zero, digits of U uniform in [-128, 127]
  const int s0 = seed[l & 63];
  v4i g0 = {0x01020001 ^ (s0 & 0x01010101), 0x02010100, 0x00010201, 0x01000102}, g1 = {0x02000101, 0x01010002, 0x00020100, 0x01010201};
  v4i m0 = {0, 0x00000100, 0, 0}, m1 = {0x00010000, 0, 0, 0};
  v4i ms = {0x00000001, 0, 0x00000100, 0}; // compressed mask: kept values
  v8i b8 = {s0, s0 * 3, s0 * 5, s0 * 7, s0 * 11, s0 * 13, s0 * 17, s0 * 19};
  v4i b0 = {b8[0], b8[1], b8[2], b8[3]}, b1 = {b8[4], b8[5], b8[6], b8[7]};
  v16i cg0 = {0}, cg1 = {0}, cg2 = {0}, cg3 = {0}, cm0 = {0}, cm1 = {0}, cm2 = {0}, cm3 = {0};
  const int idx = 0x4E4E4E4E;
  for (int i = 0; i < iters; ++i) {
    cg0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g0, b0, cg0, 0, 0, 0);
    cg1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g0, b1, cg1, 0, 0, 0);
    cg2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g1, b0, cg2, 0, 0, 0);
    cg3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g1, b1, cg3, 0, 0, 0);
    cg0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g1, b1, cg0, 0, 0, 0);
    cg1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g1, b0, cg1, 0, 0, 0);
    cg2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g0, b1, cg2, 0, 0, 0);
    cg3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g0, b0, cg3, 0, 0, 0);
    if (SPARSE) {
      cm0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(ms, b8, cm0, idx, 0, 0);
      cm1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(ms, b8, cm1, idx, 0, 0);
      cm2 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(ms, b8, cm2, idx, 0, 0);
      cm3 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(ms, b8, cm3, idx, 0, 0);
    } else {
      cm0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m0, b0, cm0, 0, 0, 0);
      cm1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m0, b1, cm1, 0, 0, 0);
      cm2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m1, b0, cm2, 0, 0, 0);
      cm3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m1, b1, cm3, 0, 0, 0);
      cm0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m1, b1, cm0, 0, 0, 0);
      cm1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m1, b0, cm1, 0, 0, 0);
      cm2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m0, b1, cm2, 0, 0, 0);
      cm3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(m0, b0, cm3, 0, 0, 0);
    }
  }
  int s = 0;
  for (int r = 0; r < 16; ++r) s += cg0[r] + cg1[r] + cg2[r] + cg3[r] + cm0[r] + cm1[r] + cm2[r] + cm3[r];
  if (s == 0x7fffffff) out[0] = s;
}

template <bool SPARSE> static double run(int ncu, int iters, const int *seed, int *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mix_kernel<SPARSE>, dim3(ncu), dim3(512), 0, 0, 1000, seed, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(mix_kernel<SPARSE>, dim3(ncu), dim3(512), 0, 0, iters, seed, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, iters = 400000;
  int *out, *seed, hs[64];
  for (int i = 0; i < 64; ++i) hs[i] = 0x9E3779B9u * (i + 1);
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&seed, sizeof hs);
  (void)hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice);
  const double md = run<false>(ncu, iters, seed, out), ms = run<true>(ncu, iters, seed, out);
  printf("per 64 bytes of K and wavefront: 8 dense + 8 dense (today's mix) %.3f ms; 8 dense + 4 sparse %.3f ms; ratio %.3f\n", md, ms,
         ms / md);
  const double ops = (double)ncu * 8 * iters * 16.0 * 2.0 * 32 * 32 * 32; // logical int8 ops of both products
  printf("logical rate of the pair of products: %.2f -> %.2f POP/s\n", ops / (md * 1e-3) / 1e15, ops / (ms * 1e-3) / 1e15);
  return 0;
}
