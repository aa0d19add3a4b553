// The multivariate LMM kernels for every (d, c) the fixed instances of mvlmm_kernels.hip / mvlmm_kernels_wide.hip do not cover
// (d <= MV_DMAX = 8 phenotypes, c <= MV_CMAX = 12 rows of X), and the gene-environment interaction test (two SNP rows).  Same source
// as the fixed kernels (mvlmm.hip.h, DT = CT = 0): loops over run-time extents, the small matrices in private memory (every lane a
// copy, as in the fixed form), the Newton-Raphson tables in a global-memory slab per workgroup (at the caps the 2 v x 2 v Hessian,
// its inverse and the elimination scratch alone are 3 x 72^2 doubles).  One wavefront per workgroup, workgroups stride over the
// SNPs.  Meant to be complete, not fast: the per-SNP work of the common shapes stays on the register-resident kernels.
#include "mvlmm_kernels.hip.h"

using namespace gemma_hip;

namespace gemma_hip {

__global__ __launch_bounds__(64) void mvlmm_rt_kernel(MvArgs g) {
  double *scratch = g.scratch + (size_t)blockIdx.x * (size_t)MvNrLayout(g.d, g.c).DOUBLES;
  MvRt rt;
  rt.d = g.d;
  rt.c = g.c;
  for (long s = blockIdx.x; s < g.l; s += gridDim.x) {
    if (g.UtX2) {
      mv_one_snp_gxe<MvWaveLanes>(g, s, scratch);
    } else {
      MvNr<0, 0, MvWaveLanes> nr(g, scratch, rt);
      nr.x = g.UtX + s * g.ld;
      mv_one_snp<0, 0, MvWaveLanes>(g, s, nr);
    }
  }
}

__global__ __launch_bounds__(64) void mvlmm_null_rt_kernel(MvNullArgs a) { mv_null_fit<0, 0, MvWaveLanes>(a, a.g.scratch); }

} // namespace gemma_hip

// doubles of Newton-Raphson scratch one workgroup needs
extern "C" size_t gemma_hip_mvlmm_rt_scratch_(int d, int c) { return (size_t)MvNrLayout(d, c).DOUBLES; }

// g->d, g->c, g->scratch (grid x gemma_hip_mvlmm_rt_scratch_(d, c) doubles) set by the caller; returns 0 or a hipError_t
extern "C" int gemma_hip_mvlmm_launch_rt_(const MvArgs *g, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL(mvlmm_rt_kernel, dim3(grid), dim3(64), 0, s, *g);
  return (int)hipGetLastError();
}
extern "C" int gemma_hip_mvlmm_null_launch_rt_(const MvNullArgs *a, hipStream_t s) {
  hipLaunchKernelGGL(mvlmm_null_rt_kernel, dim3(1), dim3(64), 0, s, *a);
  return (int)hipGetLastError();
}
