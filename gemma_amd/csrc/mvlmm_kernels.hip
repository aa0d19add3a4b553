// Kernels of the multivariate LMM stage (mvlmm.hip.h), compiled as their own translation units so that the 48 template
// instances build in parallel with gemma_hip.hip: this one holds up to three covariates (30 kernels), mvlmm_kernels_wide.hip
// four to six covariates for up to three phenotypes (18 kernels; 8 minutes as one unit, 4.5 as two side by side).  The two launchers below are internal to libgemma_hip.so (called from
// gemma_hip.hip; not part of include/gemma_hip.h).
#include "mvlmm_kernels.hip.h"

using namespace gemma_hip;

#define MV_FOR_D(F, C) F(1, C) F(2, C) F(3, C) F(4, C) F(5, C)

extern "C" int gemma_hip_mvlmm_launch_wide_(const MvArgs *g, int d, int c, hipStream_t s);      // mvlmm_kernels_wide.hip
extern "C" int gemma_hip_mvlmm_launch_d6_(const MvArgs *g, int c, hipStream_t s);               // mvlmm_kernels_d6.hip
extern "C" int gemma_hip_mvlmm_launch_d7_(const MvArgs *g, int c, hipStream_t s);               // mvlmm_kernels_d7.hip
extern "C" int gemma_hip_mvlmm_launch_d8_(const MvArgs *g, int c, hipStream_t s);               // mvlmm_kernels_d8.hip (c = 2 only)
extern "C" int gemma_hip_mvlmm_null_launch_wide_(const MvNullArgs *a, int d, int c, hipStream_t s);

// c = covariates + 1 (the SNP row).  Returns 0, a hipError_t, or -1 for an unsupported (d, c).
extern "C" int gemma_hip_mvlmm_launch_(const MvArgs *g, int d, int c, hipStream_t s) {
  if (d == 6) return gemma_hip_mvlmm_launch_d6_(g, c, s); // six / seven phenotypes, up to three covariates: fixed kernels with two / one
  if (d == 7) return gemma_hip_mvlmm_launch_d7_(g, c, s); // wavefronts per workgroup (round 5)
  if (d == 8) return gemma_hip_mvlmm_launch_d8_(g, c, s); // one covariate only (156.8 KB of tables)
  const unsigned grid = (unsigned)((g->l + 3) / 4);
#define MV_CASE(DD, CC)                                                                        \
  if (d == DD && c == CC) {                                                                    \
    hipLaunchKernelGGL((mvlmm_kernel<DD, CC>), dim3(grid), dim3(256), 0, s, *g);               \
    return (int)hipGetLastError();                                                             \
  }
  MV_FOR_D(MV_CASE, 2)
  MV_FOR_D(MV_CASE, 3)
  MV_FOR_D(MV_CASE, 4)
#undef MV_CASE
  if (c >= 5) return gemma_hip_mvlmm_launch_wide_(g, d, c, s); // four to six covariates, up to three phenotypes
  return -1;
}

// c = covariates of the null model (>= 1)
extern "C" int gemma_hip_mvlmm_null_launch_(const MvNullArgs *a, int d, int c, hipStream_t s) {
#define MV_CASE(DD, CC)                                                                        \
  if (d == DD && c == CC) {                                                                    \
    hipLaunchKernelGGL((mvlmm_null_kernel<DD, CC>), dim3(1), dim3(64), 0, s, *a);              \
    return (int)hipGetLastError();                                                             \
  }
  MV_FOR_D(MV_CASE, 1)
  MV_FOR_D(MV_CASE, 2)
  MV_FOR_D(MV_CASE, 3)
#undef MV_CASE
  if (c >= 4) return gemma_hip_mvlmm_null_launch_wide_(a, d, c, s);
  return -1;
}
