// Which XCD does workgroup b land on, for 256- and 512-thread workgroups (with the GEMM's LDS footprint)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  __shared__ double lds[9216];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = (int)(x & 0xf) + (lds[5] > 1e30 ? 1 : 0);
  }
}
int main() {
  int *d, h[64];
  hipMalloc(&d, 4096 * 4);
  for (int nt : {256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(4096), dim3(nt), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads=%4d:", nt);
    for (int i = 0; i < 40; ++i) printf(" %d", h[i]);
    printf("\n");
  }
  return 0;
}
