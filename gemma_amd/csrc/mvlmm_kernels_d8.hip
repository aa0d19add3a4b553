// FIXED kernel of the multivariate LMM stage for EIGHT phenotypes with one covariate (round 5; see mvlmm_kernels_d6.hip).  The
// Newton-Raphson tables of (d = 8, c = 2) are 156.8 of the CU's 160 KiB of LDS: one wavefront per workgroup; with more covariates they
// do not fit and the shape stays on the run-time kernel.
#include "mvlmm_kernels.hip.h"

using namespace gemma_hip;

// c = covariates + 1 (the SNP row).  Returns 0, a hipError_t, or -1 for an unsupported c.
extern "C" int gemma_hip_mvlmm_launch_d8_(const MvArgs *g, int c, hipStream_t s) {
#define MV_CASE(DD, CC, WV)                                                                                       \
  if (c == CC) {                                                                                                  \
    static_assert((size_t)WV * MvNrScratch<DD, CC>::DOUBLES * 8 <= 160 * 1024, "LDS");                            \
    hipLaunchKernelGGL((mvlmm_kernel_w<DD, CC, WV>), dim3((unsigned)((g->l + WV - 1) / WV)), dim3(64 * WV), 0, s, *g); \
    return (int)hipGetLastError();                                                                                \
  }
  MV_CASE(8, 2, 1)
#undef MV_CASE
  return -1;
}
