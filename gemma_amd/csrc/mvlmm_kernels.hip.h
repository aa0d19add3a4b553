// The two kernel templates of the multivariate LMM stage; included by the translation units that instantiate them
// (mvlmm_kernels.hip: up to three covariates; mvlmm_kernels_wide.hip: four to six covariates for up to three phenotypes;
// mvlmm_kernels_d6.hip / _d7.hip: six / seven phenotypes with up to three covariates).
#pragma once
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

// the same with WAVES wavefronts per workgroup: six / seven phenotypes, whose Newton-Raphson tables (56-110 KB per SNP) leave room for
// two / one (mvlmm_kernels_d6.hip, mvlmm_kernels_d7.hip)
template <int D, int C, int WAVES> __global__ __launch_bounds__(64 * WAVES) void mvlmm_kernel_w(MvArgs g) {
  __shared__ double scratch[WAVES][MvNrScratch<D, C>::DOUBLES];
  const int wv = (int)(threadIdx.x >> 6);
  const long s = (long)blockIdx.x * WAVES + wv;
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
