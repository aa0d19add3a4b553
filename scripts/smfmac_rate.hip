// Issue rate of v_smfmac_i32_32x32x64_i8 (2:4 sparse A) against v_mfma_i32_32x32x32_i8 (dense) on gfx950: four independent
// accumulator chains per wavefront, 4 wavefronts per CU on every CU, register operands only.  Companion of
// scripts/smfmac_probe.hip (groundwork for DESIGN.md 8, item 1).
//   hipcc --offload-arch=gfx950 -O2 scripts/smfmac_rate.hip -o /tmp/smfmac_rate && /tmp/smfmac_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <bool SPARSE>
__global__ __launch_bounds__(256) void rate_kernel(int iters, int *out) {
  const int l = threadIdx.x;
  v4i a = {l, l * 3, l * 5, l * 7};
  v8i b8 = {l, l + 1, l + 2, l + 3, l + 4, l + 5, l + 6, l + 7};
  v4i b4 = {l, l + 1, l + 2, l + 3};
  v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const int idx = 0x4E4E4E4E;
  for (int i = 0; i < iters; ++i) {
    if (SPARSE) {
      c0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b8, c0, idx, 0, 0);
      c1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b8, c1, idx, 0, 0);
      c2 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b8, c2, idx, 0, 0);
      c3 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b8, c3, idx, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b4, c3, 0, 0, 0);
    }
  }
  int s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 0x7fffffff) out[0] = s;
}

template <bool SPARSE> static double run(int ncu, int iters, int *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<SPARSE>, dim3(ncu), dim3(256), 0, 0, 100, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<SPARSE>, dim3(ncu), dim3(256), 0, 0, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, iters = 200000;
  int *out;
  hipMalloc(&out, 4);
  const double md = run<false>(ncu, iters, out), ms = run<true>(ncu, iters, out);
  const double n_inst = (double)ncu * 4 * 4 * iters; // wavefront-level instructions
  printf("CUs %d, %d iterations x 4 chains x 4 wavefronts per CU\n", ncu, iters);
  printf("dense  v_mfma_i32_32x32x32_i8 : %.3f ms, %.1f clk-equivalents per instruction per SIMD at 2.4 GHz, %.2f POP/s (2 x 32 x 32 x 32)\n",
         md, md * 1e-3 * 2.4e9 / (4.0 * iters), n_inst * 2.0 * 32 * 32 * 32 / (md * 1e-3) / 1e15);
  printf("sparse v_smfmac_i32_32x32x64_i8: %.3f ms, %.1f clk-equivalents per instruction per SIMD at 2.4 GHz, %.2f POP/s logical (2 x 32 x 32 x 64)\n",
         ms, ms * 1e-3 * 2.4e9 / (4.0 * iters), n_inst * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e15);
  return 0;
}
