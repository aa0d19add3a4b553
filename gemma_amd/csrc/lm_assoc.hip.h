// Linear model without a random effect (`gemma -lm 1..4`): LM::AnalyzeBimbam / AnalyzePlink,
// GEMMA src/lm.cpp:382-640 with CalcvPv (:224-263) and LmCalcP (:266-287).  Same ingest as the LMM path
// (2-bit decode, drop, mean imputation), no kinship, no rotation: per SNP x'x, x'y, W'x (one wavefront per
// SNP, c+1 coalesced passes over the imputed row), then the c x c projections and the three tests.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "lmm_assoc.hip.h"

namespace gemma_hip {

constexpr int LM_CMAX = 16;

struct LmArgs {
  const double *X; // l x ld, SNP-major, imputed
  long ld, l;
  const double *Wt;   // c x n
  const double *y;    // n
  const double *WtWi; // c x c
  const double *Wty;  // c
  double yPwy;
  int n, c, test_mode; // test_mode = a_mode - 50
  double lnbeta_half_df;
  SumStat *out;
};

__global__ __launch_bounds__(256) void lm_assoc_kernel(LmArgs g) {
  __shared__ double swtx[4][LM_CMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s = (long)blockIdx.x * 4 + wave;
  if (s >= g.l) return;
  const double *__restrict__ x = g.X + s * g.ld;
  const int n = g.n, c = g.c;
  double xx = 0.0, xy = 0.0;
  for (int i = lane; i < n; i += 64) {
    const double v = x[i];
    xx += v * v;
    xy += v * g.y[i];
  }
  xx = uniform(wave_sum(xx));
  xy = uniform(wave_sum(xy));
  for (int a = 0; a < c; ++a) {
    double acc = 0.0;
    const double *__restrict__ w = g.Wt + (long)a * n;
    for (int i = lane; i < n; i += 64) acc += w[i] * x[i];
    acc = wave_sum(acc);
    if (lane == 0) swtx[wave][a] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  // CalcvPv, src/lm.cpp:224-245
  double d1 = 0.0, d2 = 0.0;
  for (int a = 0; a < c; ++a) {
    double t = 0.0;
    for (int b = 0; b < c; ++b) t += g.WtWi[a * c + b] * swtx[wave][b];
    d1 += t * swtx[wave][a];
    d2 += t * g.Wty[a];
  }
  const double xPwx = xx - d1, xPwy = xy - d2, yPwy = g.yPwy;
  // LmCalcP, src/lm.cpp:266-287
  const double df = (double)n - (double)c - 1.0;
  const double yPxy = yPwy - xPwy * xPwy / xPwx;
  const double beta = xPwy / xPwx;
  const double se_wald = sqrt(yPxy / (df * xPwx));
  const double se_score = sqrt(yPwy / ((double)n * xPwx));
  const double p_wald = fdist_Q1_dev(uniform(beta * beta / (se_wald * se_wald)), df, g.lnbeta_half_df);
  const double p_score = fdist_Q1_dev(uniform(beta * beta / (se_score * se_score)), df, g.lnbeta_half_df);
  const double xl = uniform((double)n * (log(yPwy) - log(yPxy)));
  const double p_lrt = isnan(xl) ? NAN : chisq_Q1_dev(xl);
  if (lane == 0) {
    SumStat o;
    o.beta = beta;
    o.se = (g.test_mode == 3) ? se_score : se_wald;
    o.lambda_remle = 0.0;
    o.lambda_mle = 0.0;
    o.p_wald = p_wald;
    o.p_lrt = p_lrt;
    o.p_score = p_score;
    o.logl_H1 = -0.0;
    g.out[s] = o;
  }
}

} // namespace gemma_hip
