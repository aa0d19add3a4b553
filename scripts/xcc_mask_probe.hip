// How does the dispatcher hand workgroups to XCDs when the stream's CU mask leaves one XCD out (hipExtStreamCreateWithCUMask)?
// The records kernel's raster (i8gemm_sparse2.hip.h: s2_build_raster) assumes workgroup b -> XCD b % 8 on an unmasked stream; the
// two-block pipeline of round 5 runs the product on 7 XCDs.  Prints, per mask, the XCD of the first 42 workgroups, the number of
// workgroups per XCD, and whether b -> enabled[b % 7] holds for all of them.  Mask bit c: CU (c / 8) of XCD (c % 8) -- checked here too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(int *out) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned x, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = (int)(x & 0xf) + (lds[5] > 1e30 ? 1 : 0);
    out[2 * blockIdx.x + 1] = (int)hw;
    // hold the CU for a while so that later workgroups spread out as in a long kernel
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
  }
}
int main() {
  const int N = 7 * 32 * 6;
  int *d;
  hipMalloc(&d, N * 8);
  std::vector<int> h(2 * N);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  // which CUs does a mask leave?  distinct (XCC, HW_ID bits 8..15 = CU / shader array / shader engine) over all workgroups
  for (int kind = 0; kind < 5; ++kind) {
    unsigned mm[8];
    for (int w = 0; w < 8; ++w) mm[w] = 0xFFFFFFFFu;
    const char *what[5] = {"no bit cleared", "every 4th bit cleared (64)", "bits 0..31 cleared", "bits 0..127 cleared", "every 2nd bit cleared (128)"};
    for (int c = 0; c < 256; ++c) {
      const bool clr = (kind == 1 && c % 4 == 0) || (kind == 2 && c < 32) || (kind == 3 && c < 128) || (kind == 4 && c % 2 == 0);
      if (clr) mm[c >> 5] &= ~(1u << (c & 31));
    }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mm) != hipSuccess) { printf("cannot create masked stream\n"); return 1; }
    hipMemsetAsync(d, 0xFF, N * 8, s);
    hipLaunchKernelGGL(k, dim3(N), dim3(512), 128 * 1024, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
    std::vector<int> seen;
    int perx[8] = {0};
    for (int b = 0; b < N; ++b) {
      const int key = ((h[2 * b] & 15) << 16) | (h[2 * b + 1] & 0xFF00);
      bool f = false;
      for (int v : seen) f = f || v == key;
      if (!f) { seen.push_back(key); perx[h[2 * b] & 7]++; }
    }
    printf("%-34s: %zu distinct CUs hold the %d workgroups; per XCD:", what[kind], seen.size(), N);
    for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
    printf("\n");
    hipStreamDestroy(s);
  }
  for (int off = -1; off < 8; off += (off < 0 ? 1 : 7)) { // no mask, XCD 0 out, XCD 7 out
    unsigned mm[8];
    for (int w = 0; w < 8; ++w) mm[w] = 0xFFFFFFFFu;
    if (off >= 0)
      for (int c = 0; c < 32; ++c) { const int bit = 8 * c + off; mm[bit >> 5] &= ~(1u << (bit & 31)); }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mm) != hipSuccess) { printf("cannot create masked stream\n"); return 1; }
    hipMemsetAsync(d, 0xFF, N * 8, s);
    hipLaunchKernelGGL(k, dim3(N), dim3(512), 128 * 1024, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int b = 0; b < N; ++b) cnt[h[2 * b] & 15]++;
    printf("mask leaves XCD %d out: first 42 workgroups ->", off);
    for (int b = 0; b < 42; ++b) printf(" %d", h[2 * b]);
    printf("\n  workgroups per XCD:");
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    int en[8], ne = 0;
    for (int x = 0; x < 8; ++x) if (x != off) en[ne++] = x;
    int bad = 0;
    for (int b = 0; b < N; ++b) bad += h[2 * b] != en[b % ne];
    printf("\n  b -> enabled[b %% %d] violated by %d of %d workgroups\n", ne, bad, N);
    hipStreamDestroy(s);
  }
  return 0;
}
