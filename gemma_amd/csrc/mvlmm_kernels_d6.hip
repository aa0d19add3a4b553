// FIXED kernels of the multivariate LMM stage for SIX phenotypes (round 5).  Rounds 1-4 sent every shape beyond five phenotypes to the
// run-time kernel (mvlmm_kernels_rt.hip: arrays sized by the caps 8 x 12 in private memory, 60 KB per lane -- 8.6 k SNPs/s with six
// traits).  The Newton-Raphson tables of one SNP are 56-66 KB (d = 6) and 97-110 KB (d = 7) of LDS, so the fixed form fits with TWO
// wavefronts per workgroup for d = 6 and ONE for d = 7 (the kernels of mvlmm_kernels.hip.h assume four); with the extents as template
// arguments the small matrices are registers again (2.3-4.3 KB of private memory per lane).  Up to three covariates (c = 2 .. 4 rows
// of X with the SNP); everything else, and the null fit (one launch per run), stays on the run-time kernel.  One translation unit per
// d: 4 minutes each, side by side.
#include "mvlmm_kernels.hip.h"

using namespace gemma_hip;

// c = covariates + 1 (the SNP row).  Returns 0, a hipError_t, or -1 for an unsupported c.
extern "C" int gemma_hip_mvlmm_launch_d6_(const MvArgs *g, int c, hipStream_t s) {
#define MV_CASE(DD, CC, WV)                                                                                       \
  if (c == CC) {                                                                                                  \
    static_assert((size_t)WV * MvNrScratch<DD, CC>::DOUBLES * 8 <= 160 * 1024, "LDS");                            \
    hipLaunchKernelGGL((mvlmm_kernel_w<DD, CC, WV>), dim3((unsigned)((g->l + WV - 1) / WV)), dim3(64 * WV), 0, s, *g); \
    return (int)hipGetLastError();                                                                                \
  }
  MV_CASE(6, 2, 2)
  MV_CASE(6, 3, 2)
  MV_CASE(6, 4, 2)
#undef MV_CASE
  return -1;
}
