// Kernels of the multivariate LMM stage (mvlmm.hip.h), compiled as their own translation unit so that the 25 template
// instances build in parallel with gemma_hip.hip.  The two launchers below are internal to libgemma_hip.so (called from
// gemma_hip.hip; not part of include/gemma_hip.h).
#include <hip/hip_runtime.h>

#include "mvlmm.hip.h"

namespace gemma_hip {

// one SNP per wavefront, four per workgroup; the waves never synchronise with each other
template <int D, int C> __global__ __launch_bounds__(256) void mvlmm_kernel(MvArgs g) {
  __shared__ double scratch[4][MvNrScratch<D, C>::DOUBLES];
  const int wv = (int)(threadIdx.x >> 6);
  const long s = (long)blockIdx.x * 4 + wv;
  if (s >= g.l) return;
  MvNr<D, C, MvWaveLanes> nr{g, scratch[wv]};
  nr.x = g.UtX + s * g.ld;
  mv_one_snp<D, C, MvWaveLanes>(g, s, nr);
}

template <int D, int C> __global__ __launch_bounds__(64) void mvlmm_null_kernel(MvNullArgs a) {
  __shared__ double scratch[MvNrScratch<D, C>::DOUBLES];
  mv_null_fit<D, C, MvWaveLanes>(a, scratch);
}

} // namespace gemma_hip

using namespace gemma_hip;

#define MV_FOR_D(F, C) F(1, C) F(2, C) F(3, C) F(4, C) F(5, C)

// c = covariates + 1 (the SNP row).  Returns 0, a hipError_t, or -1 for an unsupported (d, c).
extern "C" int gemma_hip_mvlmm_launch_(const MvArgs *g, int d, int c, hipStream_t s) {
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
  return -1;
}
