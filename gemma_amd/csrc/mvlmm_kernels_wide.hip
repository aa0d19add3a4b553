// The multivariate LMM kernels for four to six covariates (d <= 3: the scratch of more phenotypes would not fit four
// wavefronts per workgroup); see mvlmm_kernels.hip.
#include "mvlmm_kernels.hip.h"

using namespace gemma_hip;

#define MV_FOR_D3(F, C) F(1, C) F(2, C) F(3, C)

// c = covariates + 1 (the SNP row): 5 .. 7
extern "C" int gemma_hip_mvlmm_launch_wide_(const MvArgs *g, int d, int c, hipStream_t s) {
  const unsigned grid = (unsigned)((g->l + 3) / 4);
#define MV_CASE(DD, CC)                                                                        \
  if (d == DD && c == CC) {                                                                    \
    hipLaunchKernelGGL((mvlmm_kernel<DD, CC>), dim3(grid), dim3(256), 0, s, *g);               \
    return (int)hipGetLastError();                                                             \
  }
  MV_FOR_D3(MV_CASE, 5)
  MV_FOR_D3(MV_CASE, 6)
  MV_FOR_D3(MV_CASE, 7)
#undef MV_CASE
  return -1;
}

// c = covariates of the null model: 4 .. 6
extern "C" int gemma_hip_mvlmm_null_launch_wide_(const MvNullArgs *a, int d, int c, hipStream_t s) {
#define MV_CASE(DD, CC)                                                                        \
  if (d == DD && c == CC) {                                                                    \
    hipLaunchKernelGGL((mvlmm_null_kernel<DD, CC>), dim3(1), dim3(64), 0, s, *a);              \
    return (int)hipGetLastError();                                                             \
  }
  MV_FOR_D3(MV_CASE, 4)
  MV_FOR_D3(MV_CASE, 5)
  MV_FOR_D3(MV_CASE, 6)
#undef MV_CASE
  return -1;
}
