// Standalone harness around i8gemm_packed_kernel_t (gemma_amd/csrc/i8gemm.hip.h): random packed genotypes / masks and random
// digit planes on the device, the kernel timed with HIP events, sampled output entries checked against integer sums on the host.
// Seconds per run instead of the minute a Python session costs -- the development loop for kernel variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igemma_amd/csrc scripts/i8_kernel_bench.hip -o /tmp/i8_kernel_bench
//   /tmp/i8_kernel_bench [n] [B] [variant] [gm]    variant 0 = dense mask product, 1 = sparse mask operand (i8gemm_sparse.hip.h),
//                                                  2 = records + 256 x 128 tiles (i8gemm_sparse2.hip.h), 3 = the same with
//                                                  wavefronts 8 x 1 instead of 4 x 2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "i8gemm.hip.h"
#include "smi_sampler.hpp" // scripts/: socket power / clock / power-limit residency over the timed loop (SMI=1)
#if __has_include("i8gemm_sparse.hip.h")
#include "i8gemm_sparse.hip.h"
#define HAVE_SPARSE 1
#else
#define HAVE_SPARSE 0
#endif
#if __has_include("i8gemm_sparse2.hip.h")
#include "i8gemm_sparse2.hip.h"
#define HAVE_SPARSE2 1
#else
#define HAVE_SPARSE2 0
#endif

#include "i8gemm_dense2_proto.hip.h" // scripts/: an experiment, not part of the library
#include "i8gemm_sparse2_g16_proto.hip.h" // scripts/: the records kernel with its genotype product on 16x16x64 (variant 6)
#include "i8gemm_sparse2_r16.hip.h" // the shipped 16-row kernel (variant 7)
#include "i8gemm_dense16.hip.h" // the dense byte-plane product on v_mfma_i32_16x16x64_i8 (variant 10; 11 = its genotype-masked form)
#include "i8gemm_dense16w_proto.hip.h" // scripts/: the same with 128 x 128 per wavefront, 256 x 256 x 64 tiles (variant 13; 14 = genotype-masked): slower, dropped
#include "i8gemm_sparse2_r16_persist_proto.hip.h" // scripts/: persistent workgroups / spread operand preparation (variants 8, 9)

using namespace gemma_hip;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// packed bytes g | m << 4: 1 % missing, genotypes 0/1/2; columns >= n and rows >= l are zero
__global__ void fill_A(int8_t *A, long lpad, long ldk, long l, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= lpad * ldk) return;
  const long r = i / ldk, c = i % ldk;
  int8_t v = 0;
  if (r < l && c < n) {
    const unsigned h = hash32((unsigned)(i * 2654435761u + 12345u));
    const unsigned u = h % 100;
    v = (u < 1) ? (int8_t)16 : (int8_t)((h >> 8) % 3);
  }
  A[i] = v;
}
// B_MODE (experiment: how much of the power-limited kernel time is operand toggling): 0 digits uniform in [-128, 127] (what real
// digits of U look like), 1 all zero, 2 uniform in [0, 15], 3 uniform in [-8, 7], 4 uniform in [0, 127]
__global__ void fill_B(int8_t *Bt, long total, long ldk, long n, int mode) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long c = i % ldk;
  const int hv = (int)(hash32((unsigned)(i * 40503u + 977u)) & 255);
  int v = hv - 128;
  if (mode == 1) v = 0;
  if (mode == 2) v = hv & 15;
  if (mode == 3) v = (hv & 15) - 8;
  if (mode == 4) v = hv & 127;
  if (mode == 5) v = (int)(hash32((unsigned)(i * 2654435761u + 12345u)) % 201u) - 100;
  Bt[i] = (c < n) ? (int8_t)v : (int8_t)0;
}

// stand-ins for the stages behind U^T x (CU_SPLIT experiment): grid-stride streaming kernels
__global__ __launch_bounds__(256) void side_combine_kernel(const int *__restrict__ C, size_t strideC, size_t moff, double *__restrict__ out,
                                                           size_t total) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * 1024) {
    double acc[4] = {0, 0, 0, 0};
    for (int p = 0; p < 3; ++p) {
      const int4 a = *reinterpret_cast<const int4 *>(C + p * strideC + i), b = *reinterpret_cast<const int4 *>(C + p * strideC + moff + i);
      acc[0] = acc[0] * 65536.0 + a.x + 0.5 * b.x;
      acc[1] = acc[1] * 65536.0 + a.y + 0.5 * b.y;
      acc[2] = acc[2] * 65536.0 + a.z + 0.5 * b.z;
      acc[3] = acc[3] * 65536.0 + a.w + 0.5 * b.w;
    }
    *reinterpret_cast<double4 *>(out + i) = make_double4(acc[0], acc[1], acc[2], acc[3]);
  }
}
__global__ __launch_bounds__(256) void side_read_kernel(const double *__restrict__ in, size_t total, double *__restrict__ sink) {
  double s = 0.0;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i + 1 < total; i += (size_t)gridDim.x * 512) {
    const double2 v = *reinterpret_cast<const double2 *>(in + i);
    s += v.x * v.y;
  }
  if (s == 1.2345e-300) *sink = s;
}

__global__ void count_diff_kernel(const int *a, const int *b, size_t total, unsigned long long *cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) c += a[i] != b[i];
  if (c) atomicAdd(cnt, c);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 20000, B = argc > 2 ? atol(argv[2]) : 20000;
  const int variant = argc > 3 ? atoi(argv[3]) : 0;
  // DIGITS / FUSE (variants 3, 6, 7): an odd digit count leaves the first plane with one digit; FUSE=0: one plane per digit
  const int digits = getenv("DIGITS") ? atoi(getenv("DIGITS")) : 6, fuse = getenv("FUSE") ? atoi(getenv("FUSE")) : 1;
  const int nplanes = fuse ? (digits + 1) / 2 : digits;
  const long ldk = (n + I8_BK - 1) / I8_BK * I8_BK, npad = (n + I8_BN - 1) / I8_BN * I8_BN;
  const long rowtile = variant >= 2 ? 256 : I8P_BM;
  const long lpad = (B + rowtile - 1) / rowtile * rowtile, mrows = 2 * lpad;
  const int gm = argc > 4 ? atoi(argv[4]) : 0;
  int8_t *A, *Bt;
  int *C;
  CK(hipMalloc(&A, lpad * ldk));
  CK(hipMalloc(&Bt, (size_t)digits * npad * ldk));
  CK(hipMalloc(&C, (size_t)nplanes * mrows * npad * 4));
  hipLaunchKernelGGL(fill_A, dim3((unsigned)((lpad * ldk + 255) / 256)), dim3(256), 0, 0, A, lpad, ldk, B, n);
  hipLaunchKernelGGL(fill_B, dim3((unsigned)(((size_t)digits * npad * ldk + 255) / 256)), dim3(256), 0, 0, Bt,
                     (long)digits * npad * ldk, ldk, n, getenv("B_MODE") ? atoi(getenv("B_MODE")) : 0);
  CK(hipDeviceSynchronize());
  I8PackArgs g;
  g.A = A; g.Bt = Bt; g.C = C;
  g.ldk = ldk; g.ldc = npad;
  g.strideB = npad * ldk; g.strideC = mrows * npad;
  g.m_row0 = lpad;
  g.tiles_m = (int)(lpad / I8P_BM); g.tiles_n = (int)(npad / I8_BN);
  g.nk = (int)(ldk / I8_BK);
  g.gm = argc > 4 ? atoi(argv[4]) : 0; g.fuse = fuse; g.digits = digits;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)nplanes);
#if HAVE_SPARSE
  SparseMeta sm;
  if (variant == 1 && sparse_meta_build(A, lpad, ldk, &sm)) return 1;
#endif
#if HAVE_SPARSE2
  Sparse2Args g2;
  uint4 *AM = nullptr;
  int *rsur = nullptr;
  dim3 grid2;
  if (variant >= 2) {
    const long total = lpad * (ldk / I8_BK) * 4;
    CK(hipMalloc(&AM, (size_t)total * sizeof(uint4)));
    CK(hipMalloc(&rsur, lpad * sizeof(int)));
    CK(hipMemset(rsur, 0, lpad * sizeof(int)));
    hipEvent_t m0, m1;
    CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
    CK(hipEventRecord(m0));
    hipLaunchKernelGGL(sparse2_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, A, lpad, ldk, AM, rsur);
    CK(hipEventRecord(m1));
    CK(hipEventSynchronize(m1));
    float mms = 0;
    CK(hipEventElapsedTime(&mms, m0, m1));
    printf("sparse2_meta_kernel: %.3f ms\n", mms);
    g2.AM = AM; g2.Bt = Bt; g2.C = C; g2.ldk = ldk; g2.ldc = npad; g2.strideB = npad * ldk; g2.strideC = mrows * npad;
    if (getenv("RASTER") && atoi(getenv("RASTER")) > 0) { // cross-XCD raster (s2_build_raster): row blocks of the super-patch
      std::vector<int2> map;
      s2_build_raster((int)(lpad / S2_BM), (int)(npad / S2_BN), atoi(getenv("RASTER")), map, getenv("RASTER_PR") ? atoi(getenv("RASTER_PR")) : 8);
      int2 *dmap = nullptr;
      CK(hipMalloc(&dmap, map.size() * sizeof(int2)));
      CK(hipMemcpy(dmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice));
      g2.tile_map = dmap;
    }
    g2.m_row0 = lpad; g2.tiles_m = (int)(lpad / S2_BM); g2.tiles_n = (int)(npad / S2_BN); g2.nk = (int)(ldk / I8_BK);
    g2.gm = gm; g2.fuse = fuse; g2.digits = digits;
    grid2 = dim3((unsigned)(g2.tiles_m * g2.tiles_n), (unsigned)nplanes);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_kernel_t<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           S2_NST * S2_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_kernel_t<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           S2_NST * S2_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_g16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           S2_NST * S2_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_r16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           S2_R16_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_r16p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           S2_NST * S2_STAGE));
  }
#endif
  // CU_SPLIT=k (experiment): the product runs on a stream whose CU mask leaves k CUs out, and on exactly those a streaming kernel
  // with the traffic of the stages that follow U^T x in a step (combine: 3 + 3 int32 planes in, fp64 out; then two reads of it) runs at
  // the same time.  CU_MODE=0: the k lowest mask bits, 1: every (256 / k)-th bit.
  hipStream_t mstream = 0, sstream = 0;
  const int cu_split = getenv("CU_SPLIT") ? atoi(getenv("CU_SPLIT")) : 0;
  double *side_out = nullptr;
  if (cu_split > 0) {
    const int mode = getenv("CU_MODE") ? atoi(getenv("CU_MODE")) : 0;
    unsigned mm[8], sm_[8];
    for (int w = 0; w < 8; ++w) { mm[w] = 0xFFFFFFFFu; sm_[w] = 0; }
    for (int q = 0; q < cu_split; ++q) {
      const int bit = mode == 0 ? q : q * (256 / cu_split);
      mm[bit >> 5] &= ~(1u << (bit & 31));
      sm_[bit >> 5] |= 1u << (bit & 31);
    }
    CK(hipExtStreamCreateWithCUMask(&mstream, 8, mm));
    CK(hipExtStreamCreateWithCUMask(&sstream, 8, sm_));
    CK(hipMalloc(&side_out, (size_t)lpad * npad * 8));
  }
  // variant 4: the dense byte-plane product on 256 x 256 x 64 tiles (i8gemm_dense2.hip.h): A's bytes as they are, one plane per digit;
  // variant 5: the kernel it replaces for dosages (i8gemm_packed_kernel_t<false, true>) on the same operands
  Dense2Args gd;
  gd.A = A; gd.Bt = Bt; gd.C = C; gd.ldk = ldk; gd.ldc = npad; gd.strideB = npad * ldk; gd.strideC = lpad * npad;
  gd.tiles_m = (int)(lpad / D2_BM); gd.tiles_n = (int)(npad / D2_BN); gd.nk = (int)(ldk / D2_BK); gd.gm = gm;
  if (variant == 4)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           D2_NST * D2_STAGE));
  if (variant == 5 || (variant >= 10 && variant <= 16)) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<false, true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<false, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense16_kernel_t<true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense16_kernel_t<false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense16w_kernel_t<true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, DW_NST * DW_STAGE));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense16w_kernel_t<false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, DW_NST * DW_STAGE));
  }
  if (getenv("A_MODE") && atoi(getenv("A_MODE")) == 1) { // dosage-like left factor: signed bytes uniform in [-100, 100]
    hipLaunchKernelGGL(fill_B, dim3((unsigned)((lpad * ldk + 255) / 256)), dim3(256), 0, 0, A, lpad * ldk, ldk, n, 5);
    CK(hipDeviceSynchronize());
  }
  auto launch = [&]() {
    if (variant == 4) {
      hipLaunchKernelGGL(i8gemm_dense2_kernel, dim3((unsigned)(gd.tiles_m * gd.tiles_n), (unsigned)digits), dim3(512),
                         D2_NST * D2_STAGE, mstream, gd);
      return;
    }
    if (variant == 5) {
      I8PackArgs g5 = g;
      g5.strideC = lpad * npad; g5.m_row0 = 0; g5.fuse = 0;
      hipLaunchKernelGGL((i8gemm_packed_kernel_t<false, true>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512),
                         3 * I8P_STAGE, mstream, g5);
      return;
    }
    if (variant == 13 || variant == 14) {
      DenseWArgs gw;
      gw.A = A; gw.Bt = Bt; gw.C = C; gw.ldk = ldk; gw.ldc = npad; gw.strideB = npad * ldk; gw.strideC = lpad * npad;
      gw.tiles_m = (int)(lpad / DW_BM); gw.tiles_n = (int)(npad / DW_BN); gw.nk = (int)(ldk / DW_BK); gw.gm = gm;
      if (variant == 13)
        hipLaunchKernelGGL((i8gemm_dense16w_kernel_t<true>), dim3((unsigned)(gw.tiles_m * gw.tiles_n), (unsigned)digits), dim3(256),
                           DW_NST * DW_STAGE, mstream, gw);
      else
        hipLaunchKernelGGL((i8gemm_dense16w_kernel_t<false>), dim3((unsigned)(gw.tiles_m * gw.tiles_n), (unsigned)digits), dim3(256),
                           DW_NST * DW_STAGE, mstream, gw);
      return;
    }
    if (variant == 12) { // the 32-row dense kernel on genotype-masked bytes (what G^T G of the integer kinship launched up to round 4)
      I8PackArgs g5 = g;
      g5.strideC = lpad * npad; g5.m_row0 = 0; g5.fuse = 0;
      hipLaunchKernelGGL((i8gemm_packed_kernel_t<false, false>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512),
                         3 * I8P_STAGE, mstream, g5);
      return;
    }
    if (variant == 10 || variant == 11) { // the same product on v_mfma_i32_16x16x64_i8 (10: raw bytes, 11: genotype-masked bytes)
      I8PackArgs g5 = g;
      g5.strideC = lpad * npad; g5.m_row0 = 0; g5.fuse = 0;
      if (variant == 10)
        hipLaunchKernelGGL((i8gemm_dense16_kernel_t<true>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512),
                           3 * I8P_STAGE, mstream, g5);
      else
        hipLaunchKernelGGL((i8gemm_dense16_kernel_t<false>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512),
                           3 * I8P_STAGE, mstream, g5);
      return;
    }
#if HAVE_SPARSE2
    if (variant == 6) {
      hipLaunchKernelGGL(i8gemm_sparse2_g16_kernel, grid2, dim3(512), S2_NST * S2_STAGE, mstream, g2);
      return;
    }
    if (variant == 7) {
      hipLaunchKernelGGL(i8gemm_sparse2_r16_kernel, grid2, dim3(512), S2_R16_LDS, mstream, g2);
      return;
    }
    if (variant == 8) { // the same kernel, PERSISTENT workgroups: PERSIST_WGS (default 256 = one per CU) walk the (plane, tile) list
      Sparse2ArgsP gp;
      static_cast<Sparse2Args &>(gp) = g2;
      gp.persist = 1;
      gp.nplanes = (int)grid2.y;
      const int wgs = getenv("PERSIST_WGS") ? atoi(getenv("PERSIST_WGS")) : 256;
      hipLaunchKernelGGL(i8gemm_sparse2_r16p_kernel, dim3((unsigned)wgs, 1), dim3(512), S2_NST * S2_STAGE, mstream, gp);
      return;
    }
    if (variant == 9) { // the prototype's non-persistent form (S2_R16_PREP_SPREAD decides what it differs in)
      Sparse2ArgsP gp;
      static_cast<Sparse2Args &>(gp) = g2;
      hipLaunchKernelGGL(i8gemm_sparse2_r16p_kernel, grid2, dim3(512), S2_NST * S2_STAGE, mstream, gp);
      return;
    }
    if (variant == 2) {
      hipLaunchKernelGGL(i8gemm_sparse2_kernel_t<2>, grid2, dim3(512), S2_NST * S2_STAGE, mstream, g2);
      return;
    }
    if (variant == 3) { // wavefronts 8 x 1
      hipLaunchKernelGGL(i8gemm_sparse2_kernel_t<1>, grid2, dim3(512), S2_NST * S2_STAGE, 0, g2);
      return;
    }
#endif
#if HAVE_SPARSE
    if (variant == 1) {
      hipLaunchKernelGGL(i8gemm_sparse_kernel, grid, dim3(512), 3 * SP_STAGE, 0, g, sm);
      return;
    }
#endif
    hipLaunchKernelGGL(i8gemm_packed_kernel_t<true>, grid, dim3(512), 3 * I8P_STAGE, 0, g);
  };
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<true>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
#if HAVE_SPARSE
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                         3 * SP_STAGE));
#endif
  launch();
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 3;
  hipEvent_t s0, s1;
  CK(hipEventCreate(&s0));
  CK(hipEventCreate(&s1));
  SmiSampler smi;
  const bool use_smi = getenv("SMI") && atoi(getenv("SMI")) && smi.open();
  if (use_smi) smi.start();
  CK(hipEventRecord(e0, mstream));
  if (cu_split > 0) CK(hipEventRecord(s0, sstream));
  for (int i = 0; i < reps; ++i) {
    launch();
    if (cu_split > 0) {
      const size_t tot = (size_t)lpad * npad;
      hipLaunchKernelGGL(side_combine_kernel, dim3((unsigned)(cu_split * 8)), dim3(256), 0, sstream, C, (size_t)mrows * npad, (size_t)lpad * npad,
                         side_out, tot);
      hipLaunchKernelGGL(side_read_kernel, dim3((unsigned)(cu_split * 8)), dim3(256), 0, sstream, side_out, tot, side_out + tot - 1);
      hipLaunchKernelGGL(side_read_kernel, dim3((unsigned)(cu_split * 8)), dim3(256), 0, sstream, side_out, tot, side_out + tot - 1);
    }
  }
  CK(hipEventRecord(e1, mstream));
  if (cu_split > 0) CK(hipEventRecord(s1, sstream));
  CK(hipEventSynchronize(e1));
  if (use_smi) {
    char tag[64];
    snprintf(tag, sizeof tag, "variant %d B_MODE %s", variant, getenv("B_MODE") ? getenv("B_MODE") : "0");
    smi.stop(tag);
  } else if (getenv("SMI") && atoi(getenv("SMI"))) {
    printf("[smi] unavailable on this box\n");
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  if (cu_split > 0) {
    CK(hipEventSynchronize(s1));
    float sms = 0;
    CK(hipEventElapsedTime(&sms, s0, s1));
    printf("CU_SPLIT=%d mode %s: side stream (combine-like + 2 reads, %.1f GB per step) %.3f ms per step\n", cu_split,
           getenv("CU_MODE") ? getenv("CU_MODE") : "0", ((double)nplanes * mrows * npad * 4 + 3.0 * lpad * npad * 8) / 1e9, sms / reps);
  }
  // FULLCMP=1 (variants 6, 7): EVERY entry of every plane against the shipped kernel (variant 3) run on the same operands
  if (getenv("FULLCMP") && atoi(getenv("FULLCMP")) && (variant == 6 || variant == 7 || variant == 8 || variant == 9)) {
    const size_t total = (size_t)nplanes * mrows * npad;
    int *C2 = nullptr;
    unsigned long long *dcnt = nullptr, hcnt = 0;
    CK(hipMalloc(&C2, total * 4));
    CK(hipMalloc(&dcnt, 8));
    CK(hipMemset(dcnt, 0, 8));
    CK(hipMemcpy(C2, C, total * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(C, 0xAB, total * 4));
    hipLaunchKernelGGL(i8gemm_sparse2_kernel_t<1>, grid2, dim3(512), S2_NST * S2_STAGE, 0, g2);
    CK(hipGetLastError());
    hipLaunchKernelGGL(count_diff_kernel, dim3(4096), dim3(256), 0, 0, C, C2, total, dcnt);
    CK(hipMemcpy(&hcnt, dcnt, 8, hipMemcpyDeviceToHost));
    printf("FULLCMP variant %d vs shipped kernel, digits %d fuse %d, n = %ld, B = %ld: %llu of %zu plane entries differ\n", variant, digits, fuse,
           n, B, hcnt, total);
    if (hcnt) return 3;
  }
  if (getenv("FULLCMP") && atoi(getenv("FULLCMP")) && (variant == 10 || variant == 11 || (variant >= 13 && variant <= 16))) {
    const size_t total = (size_t)digits * lpad * npad;
    int *C2 = nullptr;
    unsigned long long *dcnt = nullptr, hcnt = 0;
    CK(hipMalloc(&C2, total * 4));
    CK(hipMalloc(&dcnt, 8));
    CK(hipMemset(dcnt, 0, 8));
    CK(hipMemcpy(C2, C, total * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(C, 0xAB, total * 4));
    I8PackArgs g5 = g;
    g5.strideC = lpad * npad; g5.m_row0 = 0; g5.fuse = 0;
    if (variant == 10 || variant == 13 || variant == 15)
      hipLaunchKernelGGL((i8gemm_packed_kernel_t<false, true>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512), 3 * I8P_STAGE, 0, g5);
    else
      hipLaunchKernelGGL((i8gemm_packed_kernel_t<false, false>), dim3((unsigned)(g5.tiles_m * g5.tiles_n), (unsigned)digits), dim3(512), 3 * I8P_STAGE, 0, g5);
    CK(hipGetLastError());
    hipLaunchKernelGGL(count_diff_kernel, dim3(4096), dim3(256), 0, 0, C, C2, total, dcnt);
    CK(hipMemcpy(&hcnt, dcnt, 8, hipMemcpyDeviceToHost));
    printf("FULLCMP variant %d vs the 32-row dense kernel, digits %d, n = %ld, B = %ld: %llu of %zu plane entries differ\n", variant, digits, n, B, hcnt, total);
    if (hcnt) return 3;
  }
  if (variant == 11 || variant == 12 || variant == 14 || variant == 16) { // genotype-masked bytes: the sampled check below does not model this variant
    printf("variant %d, n = %ld, B = %ld, digits %d: %.2f ms per launch\n", variant, n, B, digits, ms / reps);
    return 0;
  }
  if (digits != 6 || fuse != 1) { // the sampled check below knows the default plane layout only
    printf("variant %d, n = %ld, B = %ld, digits %d fuse %d: %.2f ms per launch (%d planes)\n", variant, n, B, digits, fuse, ms / reps, nplanes);
    return 0;
  }
  // sampled check: rows and columns spread over the tiles, all planes
  std::vector<int8_t> hrow(ldk), hcol((size_t)digits * ldk);
  long bad = 0, checked = 0, surplus_rows = 0;
  for (int sr = 0; sr < 12; ++sr) {
    const long r = (long)((sr * 7919L + 13) % B);
    CK(hipMemcpy(hrow.data(), A + r * ldk, ldk, hipMemcpyDeviceToHost));
    // with the sparse operand a group of four with more than two missing calls keeps its first two (the surplus is the combine
    // step's job): the expected mask sum follows that rule for variant 1
    std::vector<int8_t> mrow(ldk);
    bool has_surplus = false;
    for (long k0 = 0; k0 < ldk; k0 += 4) {
      int cnt = 0;
      for (int q = 0; q < 4; ++q) {
        const int m = (hrow[k0 + q] >> 4) & 1;
        mrow[k0 + q] = (int8_t)((variant >= 1) ? (m && cnt < 2) : m);
        cnt += m;
      }
      has_surplus = has_surplus || cnt > 2;
    }
    surplus_rows += has_surplus;
    for (int sc = 0; sc < 12; ++sc) {
      const long c = (long)((sc * 104729L + 101) % n);
      for (int d = 0; d < digits; ++d)
        CK(hipMemcpy(hcol.data() + (size_t)d * ldk, Bt + (size_t)d * npad * ldk + c * ldk, ldk, hipMemcpyDeviceToHost));
      if (variant == 4 || variant == 5 || variant == 10 || variant == 13 || variant == 15) { // one plane per digit, A's bytes as signed values
        for (int d = 0; d < digits; ++d) {
          long e = 0;
          for (long k = 0; k < ldk; ++k) e += (long)hrow[k] * hcol[(size_t)d * ldk + k];
          int got;
          CK(hipMemcpy(&got, C + (size_t)d * lpad * npad + r * npad + c, 4, hipMemcpyDeviceToHost));
          bad += got != (int)e;
          ++checked;
        }
        continue;
      }
      for (int pl = 0; pl < nplanes; ++pl) {
        long eg = 0, em = 0; // fused pair: 256 * C_{2 pl + 1} + C_{2 pl}
        for (int dd = 1; dd >= 0; --dd) {
          const int d = 2 * pl + dd;
          long sg = 0, smk = 0;
          for (long k = 0; k < ldk; ++k) {
            sg += (long)(hrow[k] & 3) * hcol[(size_t)d * ldk + k];
            smk += (long)mrow[k] * hcol[(size_t)d * ldk + k];
          }
          eg = eg * 256 + sg;
          em = em * 256 + smk;
        }
        int got_g, got_m;
        CK(hipMemcpy(&got_g, C + (size_t)pl * mrows * npad + r * npad + c, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&got_m, C + (size_t)pl * mrows * npad + (lpad + r) * npad + c, 4, hipMemcpyDeviceToHost));
        bad += (got_g != (int)eg) + (got_m != (int)em);
        checked += 2;
      }
    }
  }
  printf("variant %d, n = %ld, B = %ld: %.2f ms per launch (%d planes); %ld of %ld sampled entries differ (%ld sampled rows with a "
         "surplus group)\n", variant, n, B, ms / reps, nplanes, bad, checked, surplus_rows);
  return bad ? 2 : 0;
}
