// Per-SNP association stage of the univariate LMM on gfx950: one wavefront owns one SNP.
//
// Replaces the serial loop body of batch_compute, GEMMA src/lmm.cpp:1526-1562 (BIMBAM) /
// :1853-1888 (PLINK): CalcUab (:1258), CalcRLScore (:1170), CalcLambda (:1945) with
// LogRL_/LogL_ f/dev1/dev12 (:484-1125) and CalcPab/PPab/PPPab (:283-482), CalcRLWald (:1127),
// gsl_cdf_fdist_Q / gsl_cdf_chisq_Q (:1161,1206,1553).
//
// Shape of the computation: every likelihood/derivative evaluation is a set of weighted inner
// products  S_k[a,b] = sum_i u_a[i] u_b[i] / (lambda*delta_i + 1)^k  (k = 1..3) over the n rotated
// individuals, followed by an O(c^3) scalar recursion.  A wavefront streams its SNP's U^T x row
// (coalesced, 8 B/lane; delta, U^T y, U^T W are shared by all SNPs and stay L2-resident), keeps
// all S_k in registers, butterfly-reduces across the 64 lanes, and then runs the reference's
// exact control flow (grid scan -> Brent -> Newton -> boundary checks) redundantly in every lane
// so that branches are wave-uniform.  No LDS, no barriers, no atomics: results are bit-identical
// for a SNP wherever it is scheduled (needed for the sharded == unsharded guarantee).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>

#include "lmm_search.hip.h"

namespace gemma_hip {

struct SumStat {
  double beta, se, lambda_remle, lambda_mle, p_wald, p_lrt, p_score, logl_H1;
};

constexpr int ASSOC_MAX_REGION = 64;

// outcome of one bracket's Brent + Newton polish (polish_bracket of lmm_search.hip.h) computed ahead of the per-SNP kernel
struct ChebResult {
  double l;
  int status; // PB_OK / PB_STOP / PB_FAILED; PB_OUTSIDE or CHEB_NONE: not available, polish with streaming evaluations
  int pad;
};
constexpr int CHEB_NONE = 7;

struct AssocArgs {
  const double *UtX;   // l x ld, SNP-major
  const double *UtZ;   // GXE only: l x ld rows U^T (x_s . env)
  const int *flip;     // GXE only: l flags, genotypes recoded 2 - x
  long ld;
  long l;
  const double *eval;  // n
  const double *Uty;   // n
  const double *UtWt;  // c x n (covariate-major copy of UtW)
  SumStat *out;
  int n;
  int a_mode;
  int n_region;
  int plink_nan_rule;
  double l_min, l_max;
  double l_mle_null, logl_mle_H0;
  double lnbeta_half_df;  // ln B(df/2, 1/2), df = n - c - 1 (host lgamma)
  double logdet_lmin, logdet_lmax; // sum_i log|l*delta_i + 1| at l_min / l_max (SNP independent; logdet_ends_kernel)
  int have_logdet_ends;
  // fixed-lambda table (lmm_grid.hip.h): per SNP the x-dependent sums at the grid lambdas, plus the SNP-independent
  // sums; columns of T: [q] = sum x^2 w_q,  [grid_xa0 + a * grid_nq + q] = sum x u_a w_q (a < c: U^T W column, a = c: U^T y)
  const double *grid_T; // l x grid_ld (this batch), nullptr when the table path is off
  const double *grid_F; // grid_nq x 16: pair sums among (w_1..w_c, y) in upper-triangle order, [15] = sum w_q
  int grid_ld, grid_nq, grid_xa0;
  int have_grid;
  double lam_grid[ASSOC_MAX_REGION + 1]; // l_min*exp(i*log(l_max/l_min)/n_region), host libm
  // Brackets already polished from Chebyshev-in-log(lambda) series (lmm_search.hip.h; cheb_scan_kernel / cheb_search_kernel
  // of lmm_grid.hip.h) for the grid intervals cheb_j0 .. cheb_j0 + cheb_nint - 1: cheb_slots[snp * cheb_nint + k] = the
  // SNP's slot in interval k or -1, cheb_res[(func * cheb_nint + k) * cheb_cap + slot] = what polish_bracket returned
  // (func 0: REML, 1: ML).  Everything else of this block describes the tables to those two kernels.
  const double *cheb_T;   // [k][col][slot]: series of the x-dependent sums, column-major over the interval's slots
  const double *cheb_F;   // [k][cheb_fld]: SNP-independent series
  const int *cheb_slots;
  const ChebResult *cheb_res;
  long cheb_cap;
  int cheb_ld, cheb_fld, cheb_xa0;
  int cheb_j0, cheb_nint;
  int have_cheb;
  unsigned long long cheb_qmask; // bit k: interval k is tabulated in Q form (series of sum a b delta H; lmm_search.hip.h)
  int cheb_final;          // 1: the final likelihood at lambda-hat (and with it CalcRLWald's sums) is assembled from the bracket's
                           //    series too -- the SNP's row is not read at all when its search ran on the tables
  int cheb_logdet_off;     // offset of the series of sum_i log(lambda delta_i + 1) inside an interval's cheb_F block
  const double *cheb_iv;   // [k][2] = {mid, 1 / half} of interval k in t = log(lambda)
  double cheb_mid[ASSOC_MAX_REGION], cheb_inv_half[ASSOC_MAX_REGION];
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double uniform(double v) {
  // every lane already holds the same value; tell the compiler (SGPR broadcast)
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

// 1/v for the weights H = 1/(lambda*delta + 1): hardware reciprocal + two Newton steps (<= 1-2 ulp from the
// correctly rounded quotient the reference computes; 5 instructions instead of the ~11 of an IEEE divide).
// Non-finite intermediate (v == 0, inf) falls back to the exact division so that edge semantics are IEEE's.
__device__ __forceinline__ double recip(double v) {
  double r = __builtin_amdgcn_rcp(v);
  double e = fma(-v, r, 1.0);
  r = fma(e, r, r);
  e = fma(-v, r, 1.0);
  r = fma(e, r, r);
  if (!(fabs(r) <= DBL_MAX)) r = 1.0 / v;
  return r;
}

// safe_sqrt, GEMMA src/mathfunc.cpp:122-131 (the reference's `fabs(d < 0.001)` is `d < 0.001`)
__device__ __forceinline__ double safe_sqrt_dev(double d) {
  double d1 = d;
  if (d < 0.001) d1 = fabs(d);
  if (d1 < 0.0) return NAN;
  return sqrt(d1);
}

// ------------------------------------------------------------------ special functions
// continued fraction of the regularised incomplete beta function (modified Lentz; the
// algorithm of GSL cdf/beta_inc.c beta_cont_frac: <= 512 double steps, same stopping rules)
__device__ inline double beta_cont_frac_dev(double a, double b, double x, double epsabs) {
  const double cutoff = 2.0 * DBL_MIN;
  double num = 1.0;
  double den = 1.0 - (a + b) * x / (a + 1.0);
  if (fabs(den) < cutoff) den = NAN;
  den = 1.0 / den;
  double cf = den;
  int it = 0;
  for (; it < 512; ++it) {
    const int k = it + 1;
    double coeff = k * (b - k) * x / (((a - 1.0) + 2 * k) * (a + 2 * k));
    den = 1.0 + coeff * den;
    num = 1.0 + coeff / num;
    if (fabs(den) < cutoff) den = NAN;
    if (fabs(num) < cutoff) num = NAN;
    den = 1.0 / den;
    double delta = den * num;
    cf *= delta;
    coeff = -(a + k) * (a + b + k) * x / ((a + 2 * k) * (a + 2 * k + 1.0));
    den = 1.0 + coeff * den;
    num = 1.0 + coeff / num;
    if (fabs(den) < cutoff) den = NAN;
    if (fabs(num) < cutoff) num = NAN;
    den = 1.0 / den;
    delta = den * num;
    cf *= delta;
    if (fabs(delta - 1.0) < 2.0 * DBL_EPSILON) break;
    if (cf * fabs(delta - 1.0) < epsabs) break;
  }
  if (it >= 512) return NAN;
  return cf;
}

// A*I_x(a,b)+Y with ln B(a,b) supplied (GSL cdf/beta_inc.c beta_inc_AXPY). On this path one of
// (a,b) is always 1/2 (nu1 = 1), which is what the two asymptotic branches assume.
__device__ inline double beta_inc_axpy_dev(double A, double Y, double a, double b, double x,
                                           double ln_beta) {
  if (x == 0.0) return A * 0 + Y;
  if (x == 1.0) return A * 1 + Y;
  if (a > 1e5 && b < 10 && x > a / (a + b) && b == 0.5) {
    const double N = a + (b - 1.0) / 2.0;
    return A * erfc(sqrt(-N * log(x))) + Y; // Q(1/2, z) = erfc(sqrt z)
  }
  if (b > 1e5 && a < 10 && x < b / (a + b) && a == 0.5) {
    const double N = b + (a - 1.0) / 2.0;
    return A * erf(sqrt(-N * log1p(-x))) + Y; // P(1/2, z) = erf(sqrt z)
  }
  const double ln_pre = -ln_beta + a * log(x) + b * log1p(-x);
  const double prefactor = exp(ln_pre);
  if (x < (a + 1.0) / (a + b + 2.0)) {
    const double epsabs = fabs(Y / (A * prefactor / a)) * DBL_EPSILON;
    const double cf = beta_cont_frac_dev(a, b, x, epsabs);
    return A * (prefactor * cf / a) + Y;
  } else {
    const double epsabs = fabs((A + Y) / (A * prefactor / b)) * DBL_EPSILON;
    const double cf = beta_cont_frac_dev(b, a, 1.0 - x, epsabs);
    const double term = prefactor * cf / b;
    if (A == -Y) return -A * term;
    return A * (1 - term) + Y;
  }
}

// gsl_cdf_fdist_Q(x, 1, df)  (GSL cdf/fdist.c), ln B(1/2, df/2) precomputed on the host
__device__ inline double fdist_Q1_dev(double x, double df, double lnbeta_half_df) {
  const double r = df; // nu2/nu1
  if (x < r) {
    const double u = x / (r + x);
    return beta_inc_axpy_dev(-1.0, 1.0, 0.5, df / 2.0, u, lnbeta_half_df);
  } else {
    const double u = r / (r + x);
    return beta_inc_axpy_dev(1.0, 0.0, df / 2.0, 0.5, u, lnbeta_half_df);
  }
}

// gsl_cdf_chisq_Q(x, 1) = gsl_cdf_gamma_Q(x, 1/2, 2)  (GSL cdf/gamma.c)
__device__ inline double chisq_Q1_dev(double x) {
  if (x <= 0.0) return 1.0;
  const double y = x / 2.0;
  if (y < 0.5) return 1.0 - erf(sqrt(y));
  return erfc(sqrt(y));
}

// ------------------------------------------------------------------ row-0 sums
template <int C>
struct Row0 {
  static constexpr int NI = (C + 3) * (C + 2) / 2;
  double s1[NI], s2[NI], s3[NI];
  double tr1, tr2, logdet;
};

// One pass over the SNP's rotated genotype row.  ORDER = highest power of H needed (1..3);
// ORDER = 0 means H == 1 (the "Iab" call of LogRL_f, src/lmm.cpp:839-840).
template <int C, int ORDER, bool LOGDET, int UNR = 2, bool WL = false>
__device__ __forceinline__ void row0_pass(const AssocArgs &g, const double *__restrict__ x,
                                          const double *__restrict__ y, const double *__restrict__ wl, double lambda,
                                          int lane, Row0<C> &R) {
  constexpr int NI = Row0<C>::NI;
#pragma unroll
  for (int q = 0; q < NI; ++q) R.s1[q] = R.s2[q] = R.s3[q] = 0.0;
  double tr1 = 0.0, tr2 = 0.0, ld = 0.0;
  const int n = g.n;
#pragma unroll UNR
  for (int i = lane; i < n; i += 64) {
    double u[C + 2];
#pragma unroll
    for (int a = 0; a < C; ++a) u[a] = (WL && a == C - 1) ? wl[i] : g.UtWt[(long)a * n + i];
    u[C] = x[i];
    u[C + 1] = y[i];
    double h1 = 1.0, h2 = 1.0, h3 = 1.0;
    if (ORDER >= 1) {
      const double v = g.eval[i] * lambda + 1.0;
      h1 = recip(v);
      if (ORDER >= 2) h2 = h1 * h1;
      if (ORDER >= 3) h3 = h2 * h1;
      if (LOGDET) ld += log(fabs(v));
      tr1 += h1;
      if (ORDER >= 3) tr2 += h2;
    }
#pragma unroll
    for (int a = 1; a <= C + 2; ++a) {
#pragma unroll
      for (int b = a; b <= C + 2; ++b) {
        constexpr int dummy = 0;
        (void)dummy;
        const int q = ab_index<C>(a, b);
        const double pr = u[b - 1] * u[a - 1];
        R.s1[q] += h1 * pr;
        if (ORDER >= 2) R.s2[q] += h2 * pr;
        if (ORDER >= 3) R.s3[q] += h3 * pr;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    R.s1[q] = wave_sum(R.s1[q]);
    if (ORDER >= 2) R.s2[q] = wave_sum(R.s2[q]);
    if (ORDER >= 3) R.s3[q] = wave_sum(R.s3[q]);
  }
  R.tr1 = wave_sum(tr1);
  R.tr2 = (ORDER >= 3) ? wave_sum(tr2) : 0.0;
  R.logdet = LOGDET ? wave_sum(ld) : 0.0;
}

// The projection recursion of CalcPab / CalcPPab / CalcPPPab (src/lmm.cpp:326-349, :385-407,
// :445-474) from row-0 sums.  Only what the callers read is returned:
//   ww1[p], ww2[p], ww3[p] : P_{p}[w_{p+1} w_{p+1}] of each order (p = 0..C), the pivots
//   after projecting out the first p variables -- these are Pab(i, index_ww(i+1)) in :844-849,:920-925
//   yy[k][p] : (P^k)_p [y y] for p = C (after W) and p = C+1 (after W and x); xx, xy at p = C.
template <int C>
struct Proj {
  double ww1[C + 1], ww2[C + 1], ww3[C + 1];
  double yy1[2], yy2[2], yy3[2]; // [0]: row C, [1]: row C+1
  double xx1, xy1;               // row C, order 1
};

template <int C, int ORDER>
__device__ __forceinline__ void project(const Row0<C> &R, Proj<C> &P) {
  constexpr int NI = Row0<C>::NI;
  double p1[NI], p2[NI], p3[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    p1[q] = R.s1[q];
    p2[q] = (ORDER >= 2) ? R.s2[q] : 0.0;
    p3[q] = (ORDER >= 3) ? R.s3[q] : 0.0;
  }
  constexpr int iyy = ab_index<C>(C + 2, C + 2);
  constexpr int ixx = ab_index<C>(C + 1, C + 1);
  constexpr int ixy = ab_index<C>(C + 2, C + 1);
#pragma unroll
  for (int p = 1; p <= C + 1; ++p) {
    // state before this step = row p-1
    const int iww = ab_index<C>(p, p);
    P.ww1[p - 1] = p1[iww];
    P.ww2[p - 1] = p2[iww];
    P.ww3[p - 1] = p3[iww];
    if (p == C + 1) {
      P.yy1[0] = p1[iyy];
      P.yy2[0] = p2[iyy];
      P.yy3[0] = p3[iyy];
      P.xx1 = p1[ixx];
      P.xy1 = p1[ixy];
    }
    const double ps_ww = p1[iww], ps2_ww = p2[iww], ps3_ww = p3[iww];
    double n1[NI], n2[NI], n3[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) { n1[q] = p1[q]; n2[q] = p2[q]; n3[q] = p3[q]; }
#pragma unroll
    for (int a = p + 1; a <= C + 2; ++a) {
#pragma unroll
      for (int b = a; b <= C + 2; ++b) {
        const int iab = ab_index<C>(a, b), iaw = ab_index<C>(a, p), ibw = ab_index<C>(b, p);
        const double ps_ab = p1[iab], ps_aw = p1[iaw], ps_bw = p1[ibw];
        if (ps_ww != 0) {
          n1[iab] = ps_ab - ps_aw * ps_bw / ps_ww;
          if (ORDER >= 2) {
            const double ps2_ab = p2[iab], ps2_aw = p2[iaw], ps2_bw = p2[ibw];
            double r2 = ps2_ab + ps_aw * ps_bw * ps2_ww / (ps_ww * ps_ww);
            r2 -= (ps_aw * ps2_bw + ps_bw * ps2_aw) / ps_ww;
            n2[iab] = r2;
            if (ORDER >= 3) {
              const double ps3_ab = p3[iab], ps3_aw = p3[iaw], ps3_bw = p3[ibw];
              double r3 = ps3_ab - ps_aw * ps_bw * ps2_ww * ps2_ww / (ps_ww * ps_ww * ps_ww);
              r3 -= (ps_aw * ps3_bw + ps_bw * ps3_aw + ps2_aw * ps2_bw) / ps_ww;
              r3 += (ps_aw * ps2_bw * ps2_ww + ps_bw * ps2_aw * ps2_ww + ps_aw * ps_bw * ps3_ww) /
                    (ps_ww * ps_ww);
              n3[iab] = r3;
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) { p1[q] = n1[q]; p2[q] = n2[q]; p3[q] = n3[q]; }
  }
  P.yy1[1] = p1[iyy];
  P.yy2[1] = p2[iyy];
  P.yy3[1] = p3[iyy];
}

// ------------------------------------------------------------------ models
// What the likelihood code needs from one evaluation at a given lambda (alternative model:
// calc_null = false, nc_total = c + 1).
struct Agg {
  double tr1, tr2, logdet;      // sum H, sum H^2, sum log|lambda*delta + 1|
  double trace_P, trace_PP;     // tr1 - sum_i ww2_i/ww1_i ; tr2 + sum_i (ww2_i^2/ww1_i^2 - 2 ww3_i/ww1_i)
  double slog;                  // sum_i log(ww1_i), i < c+1    (log|W^T H^-1 W| pivots)
  double yy1_c, xx1, xy1;       // P_c[yy], P_c[xx], P_c[xy]     (row c: after the covariates)
  double yy1, yy2, yy3;         // (P^k)_{c+1}[yy], k = 1..3    (row c+1: after covariates and x)
};

// number of covariates fixed at compile time: everything in registers (c = 1..4)
// WL: the LAST covariate is a per-SNP vector (wlast) instead of row C-1 of UtWt (the GXE variants, where the SNP
// itself is a covariate and the tested variable is SNP x environment)
template <int C, int UNR = 2, bool WL = false>
struct FixedC {
  static constexpr bool HAS_GRID = !WL;
  static constexpr bool HAS_CHEB = !WL;
  static constexpr int CC = C;
  const double *wlast = nullptr;
  __device__ __forceinline__ int c() const { return C; }
  template <int ORDER>
  __device__ __forceinline__ void finish(const Row0<C> &R, Agg &A) const {
    Proj<C> P;
    project<C, (ORDER == 0 ? 1 : ORDER)>(R, P);
    A.tr1 = R.tr1;
    A.tr2 = R.tr2;
    A.logdet = R.logdet;
    double tp = R.tr1, tpp = R.tr2, sl = 0.0;
#pragma unroll
    for (int i = 0; i < C + 1; ++i) {
      if (ORDER >= 2) tp -= P.ww2[i] / P.ww1[i];
      if (ORDER >= 3) tpp += P.ww2[i] * P.ww2[i] / (P.ww1[i] * P.ww1[i]) - 2.0 * P.ww3[i] / P.ww1[i];
      if (ORDER <= 1) sl += log(P.ww1[i]);
    }
    A.trace_P = tp;
    A.trace_PP = tpp;
    A.slog = sl;
    A.yy1_c = P.yy1[0];
    A.xx1 = P.xx1;
    A.xy1 = P.xy1;
    A.yy1 = P.yy1[1];
    A.yy2 = P.yy2[1];
    A.yy3 = P.yy3[1];
  }
  template <int ORDER, bool LOGDET>
  __device__ __forceinline__ void eval(const AssocArgs &g, const double *x, const double *y, double l, int lane,
                                       Agg &A) const {
    Row0<C> R;
    row0_pass<C, ORDER, LOGDET, UNR, WL>(g, x, y, wlast, l, lane, R);
    finish<ORDER>(R, A);
  }
  // The same evaluation at grid lambda gi (gi < 0: H = 1, the Iab call) from the fixed-lambda table: the row-0
  // sums are assembled from the SNP's table row (pairs with x) and the SNP-independent sums; ORDER <= 2.
  template <int ORDER>
  __device__ __forceinline__ void eval_grid(const AssocArgs &g, const double *__restrict__ trow, int gi, Agg &A) const {
    Row0<C> R;
    const int q1 = (gi < 0) ? 0 : 1 + 2 * gi, q2 = q1 + 1;
    const double *__restrict__ F1 = g.grid_F + (long)q1 * 16;
    const double *__restrict__ F2 = g.grid_F + (long)q2 * 16;
    const int nq = g.grid_nq, xa0 = g.grid_xa0;
#pragma unroll
    for (int a = 1; a <= C + 2; ++a) {
#pragma unroll
      for (int b = a; b <= C + 2; ++b) {
        const int q = ab_index<C>(a, b);
        // variables 1..C: covariates (fixed index a-1), C+1: x, C+2: y (fixed index C)
        const bool ax = (a == C + 1), bx = (b == C + 1);
        const int fa = (a == C + 2) ? C : a - 1, fb = (b == C + 2) ? C : b - 1;
        double v1, v2 = 0.0;
        if (ax && bx) {
          v1 = trow[q1];
          if (ORDER >= 2) v2 = trow[q2];
        } else if (ax || bx) {
          const int f = ax ? fb : fa;
          v1 = trow[xa0 + f * nq + q1];
          if (ORDER >= 2) v2 = trow[xa0 + f * nq + q2];
        } else {
          // upper-triangle index of (fa, fb), fa <= fb, over C + 1 fixed variables
          const int pidx = fa * (C + 1) - fa * (fa - 1) / 2 + (fb - fa);
          v1 = F1[pidx];
          if (ORDER >= 2) v2 = F2[pidx];
        }
        R.s1[q] = v1;
        R.s2[q] = v2;
        R.s3[q] = 0.0;
      }
    }
    R.tr1 = (gi < 0) ? 0.0 : F1[15];
    R.tr2 = 0.0;
    R.logdet = 0.0;
    finish<ORDER>(R, A);
  }
  // The order-1 evaluation at a lambda inside tabulated interval kint (sv = its position in [-1, 1]) from the SNP's series
  // of that interval (slot) and the SNP-independent series -- what LogRL_f / LogL_f and CalcRLWald need at lambda-hat, without
  // a pass over the SNP's row.  Q-form intervals: S = S0 - lambda Q with S0 from the fixed-lambda table's weight-1 column.
  __device__ __forceinline__ void eval_cheb(const AssocArgs &g, const double *__restrict__ trow, int kint, int slot, double l,
                                            double sv, Agg &A) const {
    Row0<C> R;
    const double *__restrict__ snp = g.cheb_T + (long)kint * g.cheb_cap * g.cheb_ld + slot;
    const double *__restrict__ fix = g.cheb_F + (long)kint * g.cheb_fld;
    const bool qf = (g.cheb_qmask >> kint) & 1ull;
    constexpr int NPAIR = (C + 1) * (C + 2) / 2;
#pragma unroll
    for (int a = 1; a <= C + 2; ++a) {
#pragma unroll
      for (int b = a; b <= C + 2; ++b) {
        const int q = ab_index<C>(a, b);
        const bool ax = (a == C + 1), bx = (b == C + 1);
        const int fa = (a == C + 2) ? C : a - 1, fb = (b == C + 2) ? C : b - 1;
        double v, d1, d2, s0;
        if (ax && bx) {
          cheb_eval<0>(snp, g.cheb_cap, sv, v, d1, d2);
          s0 = trow[0];
        } else if (ax || bx) {
          const int f = ax ? fb : fa;
          cheb_eval<0>(snp + (long)(g.cheb_xa0 + f * CHEB_N) * g.cheb_cap, g.cheb_cap, sv, v, d1, d2);
          s0 = trow[g.grid_xa0 + f * g.grid_nq];
        } else {
          const int pidx = fa * (C + 1) - fa * (fa - 1) / 2 + (fb - fa);
          cheb_eval<0>(fix + pidx * CHEB_N, 1, sv, v, d1, d2);
          s0 = g.grid_F[pidx];
        }
        R.s1[q] = qf ? s0 - l * v : v;
        R.s2[q] = 0.0;
        R.s3[q] = 0.0;
      }
    }
    double gsum, ld, d1, d2;
    cheb_eval<0>(fix + NPAIR * CHEB_N, 1, sv, gsum, d1, d2);
    cheb_eval<0>(fix + g.cheb_logdet_off, 1, sv, ld, d1, d2);
    R.tr1 = (double)g.n - gsum; // sum H = n - sum (1 - H)
    R.tr2 = 0.0;
    R.logdet = ld;
    finish<1>(R, A);
  }
};

// any number of covariates (c <= GEN_CMAX): the (c+2) x (c+2) product table is covered by 4 x 4 register
// tiles, one streaming pass per tile pair; row-0 sums and the projection recursion live in per-wave LDS.
constexpr int GEN_CMAX = 16;
constexpr int GEN_NI = (GEN_CMAX + 3) * (GEN_CMAX + 2) / 2;
constexpr int GEN_LDS_PER_WAVE = 6 * GEN_NI;
// Beyond 16 covariates (20 principal components + age + sex is an everyday GWAS model; the reference is generic in n_cvt,
// src/lmm.cpp:283-357) the same code runs with ONE wavefront per workgroup and its six tables in dynamic LDS:
// 6 (c + 3)(c + 2) / 2 doubles = 106 KiB at c = 64.  "wide" kernels below; slow (the product table takes (c + 2)^2 / 32 passes
// per evaluation) but complete.
constexpr int GEN_CMAX_WIDE = 64;
__host__ __device__ inline int gen_ni_for(int c) { return (c + 3) * (c + 2) / 2; }

__device__ __forceinline__ int ab_index_rt(int a, int b, int c) {
  const int cols = c + 2;
  const int a1 = (b <= a) ? b : a;
  const int b1 = (b <= a) ? a : b;
  return (2 * cols - a1 + 2) * (a1 - 1) / 2 + b1 - a1;
}

struct GenericC {
  static constexpr bool HAS_GRID = false;
  static constexpr bool HAS_CHEB = false;
  static constexpr int CC = 1; // unused
  int cc;
  double *L; // this wave's LDS scratch: 6 tables of `ni` doubles
  int ni = GEN_NI; // table stride: GEN_NI in the 4-wavefront kernels, gen_ni_for(c) in the wide ones
  const double *wlast = nullptr; // non-null: the last covariate is this per-SNP vector (GXE)
  __device__ __forceinline__ int c() const { return cc; }

  template <int ORDER, bool LOGDET>
  __device__ void eval(const AssocArgs &g, const double *__restrict__ x, const double *__restrict__ y, double lambda,
                       int lane, Agg &A) const {
    const int c = cc, nv = c + 2, n = g.n;
    constexpr int EO = (ORDER == 0) ? 1 : ORDER;
    double *s1 = L, *s2 = L + ni, *s3 = L + 2 * ni;
    double tr1 = 0.0, tr2 = 0.0, ld = 0.0;
    const int nblk = (nv + 3) >> 2;
    for (int bi = 0; bi < nblk; ++bi) {
      for (int bj = bi; bj < nblk; ++bj) {
        const double *pa[4], *pb[4];
        bool va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = 4 * bi + k, w = 4 * bj + k;
          va[k] = v < nv;
          vb[k] = w < nv;
          const int v0 = va[k] ? v : 0, w0 = vb[k] ? w : 0;
          pa[k] = (v0 < c) ? ((wlast && v0 == c - 1) ? wlast : g.UtWt + (long)v0 * n) : (v0 == c ? x : y);
          pb[k] = (w0 < c) ? ((wlast && w0 == c - 1) ? wlast : g.UtWt + (long)w0 * n) : (w0 == c ? x : y);
        }
        double a1[4][4], a2[4][4], a3[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int m = 0; m < 4; ++m) a1[k][m] = a2[k][m] = a3[k][m] = 0.0;
        const bool first = (bi == 0 && bj == 0);
        for (int i = lane; i < n; i += 64) {
          double h1 = 1.0, h2 = 1.0, h3 = 1.0;
          if (ORDER >= 1) {
            const double v = g.eval[i] * lambda + 1.0;
            h1 = recip(v);
            if (ORDER >= 2) h2 = h1 * h1;
            if (ORDER >= 3) h3 = h2 * h1;
            if (first) {
              if (LOGDET) ld += log(fabs(v));
              tr1 += h1;
              if (ORDER >= 3) tr2 += h2;
            }
          }
          double ua[4], ub[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ua[k] = va[k] ? pa[k][i] : 0.0;
            ub[k] = vb[k] ? pb[k][i] : 0.0;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const double pr = ub[m] * ua[k];
              a1[k][m] += h1 * pr;
              if (ORDER >= 2) a2[k][m] += h2 * pr;
              if (ORDER >= 3) a3[k][m] += h3 * pr;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int a = 4 * bi + k + 1, b = 4 * bj + m + 1;
            const double r1 = wave_sum(a1[k][m]);
            const double r2 = (ORDER >= 2) ? wave_sum(a2[k][m]) : 0.0;
            const double r3 = (ORDER >= 3) ? wave_sum(a3[k][m]) : 0.0;
            if (a <= b && b <= nv && lane == 0) {
              const int idx = ab_index_rt(a, b, c);
              s1[idx] = r1;
              if (EO >= 2) s2[idx] = r2;
              if (EO >= 3) s3[idx] = r3;
            }
          }
      }
    }
    __builtin_amdgcn_wave_barrier();
    A.tr1 = wave_sum(tr1);
    A.tr2 = (ORDER >= 3) ? wave_sum(tr2) : 0.0;
    A.logdet = LOGDET ? wave_sum(ld) : 0.0;
    // projection recursion (src/lmm.cpp:326-349, :385-407, :445-474), one lane per (a,b) pair
    double *cur1 = s1, *cur2 = s2, *cur3 = s3;
    double *nx1 = L + 3 * ni, *nx2 = L + 4 * ni, *nx3 = L + 5 * ni;
    const int iyy = ab_index_rt(c + 2, c + 2, c), ixx = ab_index_rt(c + 1, c + 1, c),
              ixy = ab_index_rt(c + 2, c + 1, c);
    double tp = A.tr1, tpp = A.tr2, sl = 0.0;
    for (int p = 1; p <= c + 1; ++p) {
      const int iww = ab_index_rt(p, p, c);
      const double ps_ww = cur1[iww];
      const double ps2_ww = (EO >= 2) ? cur2[iww] : 0.0;
      const double ps3_ww = (EO >= 3) ? cur3[iww] : 0.0;
      if (ORDER >= 2) tp -= ps2_ww / ps_ww;
      if (ORDER >= 3) tpp += ps2_ww * ps2_ww / (ps_ww * ps_ww) - 2.0 * ps3_ww / ps_ww;
      if (ORDER <= 1) sl += log(ps_ww);
      if (p == c + 1) {
        A.yy1_c = cur1[iyy];
        A.xx1 = cur1[ixx];
        A.xy1 = cur1[ixy];
      }
      int cnt = 0;
      for (int a = p + 1; a <= nv; ++a) {
        for (int b = a; b <= nv; ++b, ++cnt) {
          if ((cnt & 63) != lane) continue;
          const int iab = ab_index_rt(a, b, c), iaw = ab_index_rt(a, p, c), ibw = ab_index_rt(b, p, c);
          const double ps_ab = cur1[iab], ps_aw = cur1[iaw], ps_bw = cur1[ibw];
          double r1 = ps_ab, r2 = 0.0, r3 = 0.0;
          if (EO >= 2) r2 = cur2[iab];
          if (EO >= 3) r3 = cur3[iab];
          if (ps_ww != 0) {
            r1 = ps_ab - ps_aw * ps_bw / ps_ww;
            if (EO >= 2) {
              const double ps2_ab = cur2[iab], ps2_aw = cur2[iaw], ps2_bw = cur2[ibw];
              r2 = ps2_ab + ps_aw * ps_bw * ps2_ww / (ps_ww * ps_ww);
              r2 -= (ps_aw * ps2_bw + ps_bw * ps2_aw) / ps_ww;
              if (EO >= 3) {
                const double ps3_ab = cur3[iab], ps3_aw = cur3[iaw], ps3_bw = cur3[ibw];
                r3 = ps3_ab - ps_aw * ps_bw * ps2_ww * ps2_ww / (ps_ww * ps_ww * ps_ww);
                r3 -= (ps_aw * ps3_bw + ps_bw * ps3_aw + ps2_aw * ps2_bw) / ps_ww;
                r3 += (ps_aw * ps2_bw * ps2_ww + ps_bw * ps2_aw * ps2_ww + ps_aw * ps_bw * ps3_ww) /
                      (ps_ww * ps_ww);
              }
            }
          }
          nx1[iab] = r1;
          if (EO >= 2) nx2[iab] = r2;
          if (EO >= 3) nx3[iab] = r3;
        }
      }
      __builtin_amdgcn_wave_barrier();
      double *t;
      t = cur1; cur1 = nx1; nx1 = t;
      t = cur2; cur2 = nx2; nx2 = t;
      t = cur3; cur3 = nx3; nx3 = t;
    }
    A.trace_P = tp;
    A.trace_PP = tpp;
    A.slog = sl;
    A.yy1 = cur1[iyy];
    A.yy2 = (EO >= 2) ? cur2[iyy] : 0.0;
    A.yy3 = (EO >= 3) ? cur3[iyy] : 0.0;
    __builtin_amdgcn_wave_barrier();
  }
};

// ------------------------------------------------------------------ likelihood pieces
template <class M>
struct SnpCtx {
  const AssocArgs *g;
  const double *x; // vector in the "x" slot (the tested variable): the SNP's U^T x row
  const double *y; // vector in the "y" slot: U^T y (AnalyzeGene: the gene's row, with x = the fixed U^T x)
  int lane;
  M m;
  double logdet_iw; // sum_i log(Iab(i, ww_{i+1})), i < c+1  (H == 1; SNP constant)
  const double *trow; // this SNP's row of the fixed-lambda table, nullptr: stream every evaluation
  const int *cslots;  // this SNP's slots in the per-interval Chebyshev tables (-1: none), nullptr: no tables
};

// LogRL_dev1 / LogRL_dev12 (src/lmm.cpp:866-943, :1035-1125) and LogL_dev1 / LogL_dev12
// (:544-640, :719-797).  ORDER 2 -> dev1 only; ORDER 3 -> dev1 and dev2.
template <class M, bool REML, int ORDER>
__device__ __forceinline__ void deriv_from(const SnpCtx<M> &s, double l, const Agg &A, double &dev1, double &dev2) {
  const double n = (double)s.g->n;
  const double P_yy = A.yy1, PP_yy = A.yy2, PPP_yy = A.yy3;
  const double yPKPy = (P_yy - PP_yy) / l;
  if (REML) {
    const double df = n - (double)s.m.c() - 1.0;
    const double trace_P = A.trace_P, trace_PP = A.trace_PP;
    const double trace_PK = (df - trace_P) / l;
    dev1 = -0.5 * trace_PK + 0.5 * df * yPKPy / P_yy;
    if (ORDER >= 3) {
      const double trace_PKPK = (df + trace_PP - 2.0 * trace_P) / (l * l);
      const double yPKPKPy = (P_yy + PPP_yy - 2.0 * PP_yy) / (l * l);
      dev2 = 0.5 * trace_PKPK - 0.5 * df * (2.0 * yPKPKPy * P_yy - yPKPy * yPKPy) / (P_yy * P_yy);
    }
  } else {
    const double trace_HiK = (n - A.tr1) / l;
    dev1 = -0.5 * trace_HiK + 0.5 * n * yPKPy / P_yy;
    if (ORDER >= 3) {
      const double trace_HiKHiK = (n + A.tr2 - 2 * A.tr1) / (l * l);
      const double yPKPKPy = (P_yy + PPP_yy - 2.0 * PP_yy) / (l * l);
      dev2 = 0.5 * trace_HiKHiK - 0.5 * n * (2.0 * yPKPKPy * P_yy - yPKPy * yPKPy) / (P_yy * P_yy);
    }
  }
  dev1 = uniform(dev1);
  if (ORDER >= 3) dev2 = uniform(dev2);
}
template <class M, bool REML, int ORDER>
__device__ __forceinline__ void deriv(const SnpCtx<M> &s, double l, double &dev1, double &dev2) {
  Agg A;
  s.m.template eval<ORDER, false>(*s.g, s.x, s.y, l, s.lane, A);
  deriv_from<M, REML, ORDER>(s, l, A, dev1, dev2);
}
// first derivative at grid lambda gi: from the fixed-lambda table when this SNP has a row in it
template <class M, bool REML>
__device__ __forceinline__ double dev1_grid(const SnpCtx<M> &s, int gi) {
  double d1, d2 = 0.0;
  const double l = s.g->lam_grid[gi];
  if constexpr (M::HAS_GRID) {
    if (s.trow) {
      Agg A;
      s.m.template eval_grid<2>(*s.g, s.trow, gi, A);
      deriv_from<M, REML, 2>(s, l, A, d1, d2);
      return d1;
    }
  }
  deriv<M, REML, 2>(s, l, d1, d2);
  return d1;
}

// LogRL_f (src/lmm.cpp:799-864) / LogL_f (:484-542)
template <class M, bool REML>
__device__ __forceinline__ double logf(const SnpCtx<M> &s, double l, Agg &A, int kint = -1) {
  // log|H| at the two interval ends does not depend on the SNP: taken from the setup kernel, which sums
  // in this kernel's own order (bit-identical to evaluating it here)
  if (s.g->have_logdet_ends && (l == s.g->l_min || l == s.g->l_max)) {
    bool done = false;
    if constexpr (M::HAS_GRID) {
      if (s.trow) { // l_min / l_max are grid points 0 / n_region
        s.m.template eval_grid<1>(*s.g, s.trow, (l == s.g->l_min) ? 0 : s.g->n_region, A);
        done = true;
      }
    }
    if (!done) s.m.template eval<1, false>(*s.g, s.x, s.y, l, s.lane, A);
    A.logdet = (l == s.g->l_min) ? s.g->logdet_lmin : s.g->logdet_lmax;
  } else {
    // log|H| = sum_i log(lambda delta_i + 1) is SNP-independent and smooth in log(lambda): inside a tabulated interval
    // it comes from its Chebyshev series (lmm_grid.hip.h; ~1e-14 of its size), which takes the one log per element out
    // of the streaming pass
    bool have_ld = false;
    double ldet = 0.0;
    if constexpr (M::HAS_CHEB) {
      if (s.cslots && kint >= 0 && kint < s.g->cheb_nint) {
        const double sv = (log(l) - s.g->cheb_iv[2 * kint]) * s.g->cheb_iv[2 * kint + 1];
        if (fabs(sv) <= 1.0) {
          double d1, d2;
          cheb_eval<0>(s.g->cheb_F + (long)kint * s.g->cheb_fld + s.g->cheb_logdet_off, 1, sv, ldet, d1, d2);
          have_ld = true;
        }
      }
    }
    bool done = false;
    if constexpr (M::HAS_CHEB) {
      if (have_ld && s.g->cheb_final && s.trow) {
        const int slot = __builtin_amdgcn_readfirstlane(s.cslots[kint]);
        if (slot >= 0) {
          const double sv = (log(l) - s.g->cheb_iv[2 * kint]) * s.g->cheb_iv[2 * kint + 1];
          s.m.eval_cheb(*s.g, s.trow, kint, slot, l, sv, A);
          done = true;
        }
      }
    }
    if (done) {
      // everything came from the series (A.logdet included)
    } else if (have_ld) {
      s.m.template eval<1, false>(*s.g, s.x, s.y, l, s.lane, A);
      A.logdet = uniform(ldet);
    } else {
      s.m.template eval<1, true>(*s.g, s.x, s.y, l, s.lane, A);
    }
  }
  const double n = (double)s.g->n;
  double P_yy = A.yy1;
  if (P_yy >= 0.0 && P_yy < 1e-8) P_yy = 1e-8; // P_YY_MIN, src/lmm.cpp:52,527,854
  double f;
  if (REML) {
    const double df = n - (double)s.m.c() - 1.0;
    const double logdet_hiw = A.slog - s.logdet_iw;
    const double cst = 0.5 * df * (log(df) - log(2 * M_PI) - 1.0);
    f = cst - 0.5 * A.logdet - 0.5 * logdet_hiw - 0.5 * df * log(P_yy);
  } else {
    const double cst = 0.5 * n * (log(n) - log(2 * M_PI) - 1.0);
    f = cst - 0.5 * A.logdet - 0.5 * n * log(P_yy);
  }
  return uniform(f);
}

// CalcRLWald (src/lmm.cpp:1127-1167) / CalcRLScore (:1170-1211) from the order-1 evaluation at their lambda
template <class M, bool SCORE>
__device__ __forceinline__ void wald_score_from(const SnpCtx<M> &s, const Agg &A, double &beta, double &se,
                                                double &pval) {
  const int df = s.g->n - s.m.c() - 1;
  const double P_yy = A.yy1_c, P_xx = A.xx1, P_xy = A.xy1, Px_yy = A.yy1;
  beta = uniform(P_xy / P_xx);
  const double tau = (double)df / Px_yy;
  se = uniform(safe_sqrt_dev(1.0 / (tau * P_xx)));
  double stat;
  if (SCORE)
    stat = (double)s.g->n * P_xy * P_xy / (P_yy * P_xx);
  else
    stat = (P_yy - Px_yy) * tau;
  stat = uniform(stat);
  pval = uniform(fdist_Q1_dev(stat, (double)df, s.g->lnbeta_half_df));
}
template <class M, bool SCORE>
__device__ __forceinline__ void wald_score(const SnpCtx<M> &s, double l, double &beta, double &se,
                                           double &pval) {
  Agg A;
  s.m.template eval<1, false>(*s.g, s.x, s.y, l, s.lane, A);
  wald_score_from<M, SCORE>(s, A, beta, se, pval);
}

// ------------------------------------------------------------------ root finders: lmm_search.hip.h
// the evaluator for polish_bracket of this kernel: every evaluation a streaming pass over the SNP's row
template <class M, bool REML>
struct StreamEvaluator {
  const SnpCtx<M> *cx;
  __device__ __forceinline__ bool dev1(double l, double &d1) {
    double d2;
    deriv<M, REML, 2>(*cx, l, d1, d2);
    return true;
  }
  __device__ __forceinline__ bool dev12(double l, double &d1, double &d2) {
    deriv<M, REML, 3>(*cx, l, d1, d2);
    return true;
  }
};
// CalcLambda, src/lmm.cpp:1945-2140.  Brackets are processed as soon as the grid scan finds
// them (the evaluations are pure, so interleaving scan and polish gives the reference's
// sequence of results); `return NaN` and `break` semantics of :2057-2060,:2087-2094 are kept.
template <class M, bool REML>
__device__ __forceinline__ void calc_lambda(const SnpCtx<M> &cx, double &lambda, double &logf_out, Agg &best) {
  const AssocArgs &g = *cx.g;
  Agg cand;
  const double l_min = g.l_min, l_max = g.l_max;
  double lam = NAN, lf = NAN;
  bool any = false, first = true, stop = false, failed = false;
  double l = 0.0, l_temp = 0.0;
  double d_lo = dev1_grid<M, REML>(cx, 0);
  for (int i = 0; i < g.n_region; ++i) {
    const double lambda_l0 = g.lam_grid[i], lambda_h0 = g.lam_grid[i + 1];
    const double d_hi = dev1_grid<M, REML>(cx, i + 1);
    const bool bracket = (d_lo * d_hi <= 0);
    if (bracket) any = true;
    if (bracket && !stop && !failed) {
      int pb = PB_OUTSIDE;
      if constexpr (M::HAS_CHEB) {
        const int k = i - g.cheb_j0;
        if (cx.cslots && k >= 0 && k < g.cheb_nint) {
          const int slot = __builtin_amdgcn_readfirstlane(cx.cslots[k]);
          if (slot >= 0) {
            // this bracket was polished from the SNP's series of the interval (cheb_search_kernel); anything but a
            // clean verdict there (an iterate left the interval, a non-finite value) is redone below, streaming
            const ChebResult r = g.cheb_res[((long)(REML ? 0 : 1) * g.cheb_nint + k) * g.cheb_cap + slot];
            const int st = __builtin_amdgcn_readfirstlane(r.status);
            if (st == PB_OK || st == PB_STOP || st == PB_FAILED) {
              pb = st;
              if (st == PB_OK) l = uniform(r.l);
            }
          }
        }
      }
      if (pb == PB_OUTSIDE) {
        StreamEvaluator<M, REML> ev;
        ev.cx = &cx;
        pb = polish_bracket(ev, lambda_l0, lambda_h0, d_lo, d_hi, l_min, l_max, l, l_temp);
      }
      if (pb == PB_STOP) {
        stop = true; // :2057-2060 leaves the bracket loop
      } else if (pb == PB_FAILED) {
        failed = true; // :2087-2094: lambda = logf = NaN, return
      } else {
        const double logf_l = logf<M, REML>(cx, l, cand, i - g.cheb_j0);
        if (first) {
          lf = logf_l; lam = l; best = cand;
        } else if (lf < logf_l) {
          lf = logf_l; lam = l; best = cand;
        }
        first = false;
      }
    }
    d_lo = d_hi;
  }
  if (failed) {
    lambda = NAN;
    logf_out = NAN;
    return;
  }
  Agg a_lo, a_hi;
  const double logf_l = logf<M, REML>(cx, l_min, a_lo);
  const double logf_h = logf<M, REML>(cx, l_max, a_hi);
  if (!any) { // :1985-2000
    if (logf_l >= logf_h) { lam = l_min; lf = logf_l; best = a_lo; } else { lam = l_max; lf = logf_h; best = a_hi; }
  } else { // :2121-2136
    if (logf_l > lf) { lam = l_min; lf = logf_l; best = a_lo; }
    if (logf_h > lf) { lam = l_max; lf = logf_h; best = a_hi; }
  }
  lambda = lam;
  logf_out = lf;
}

// sum_i log(Iab(i, ww_{i+1})): CalcPab with H == 1 (src/lmm.cpp:839-850); constant for the SNP
template <class M>
__device__ __forceinline__ double logdet_iw_of(const SnpCtx<M> &cx) {
  Agg A;
  bool done = false;
  if constexpr (M::HAS_GRID) {
    if (cx.trow) {
      cx.m.template eval_grid<0>(*cx.g, cx.trow, -1, A);
      done = true;
    }
  }
  if (!done) cx.m.template eval<0, false>(*cx.g, cx.x, cx.y, 0.0, cx.lane, A);
  return uniform(A.slog);
}

// ------------------------------------------------------------------ kernels
// the body of batch_compute's loop (src/lmm.cpp:1526-1562 / :1853-1888) for one SNP
template <class M>
__device__ __forceinline__ void assoc_one_snp(const AssocArgs &g, const M &model, long snp, int lane) {
  SnpCtx<M> cx;
  cx.g = &g;
  cx.x = g.UtX + snp * g.ld;
  cx.y = g.Uty;
  cx.lane = lane;
  cx.m = model;
  cx.logdet_iw = 0.0;
  cx.trow = (g.have_grid && g.grid_T) ? g.grid_T + snp * g.grid_ld : nullptr;
  cx.cslots = (cx.trow && g.have_cheb && g.cheb_slots) ? g.cheb_slots + snp * g.cheb_nint : nullptr;
  const int a_mode = g.a_mode;

  double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
  double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0;
  bool wald_skipped = false;

  if (a_mode == 3 || a_mode == 4 || a_mode == 9) // "3 is before 1", src/lmm.cpp:1540-1543
    wald_score<M, true>(cx, g.l_mle_null, beta, se, p_score);

  if (a_mode == 1 || a_mode == 4) {
    cx.logdet_iw = logdet_iw_of(cx);
    Agg at_remle;
    calc_lambda<M, true>(cx, lambda_remle, logl_H1, at_remle);
    if (isnan(logl_H1)) {
      // failed search: the BIMBAM loop still calls CalcRLWald(NaN) (src/lmm.cpp:1548) -> NaN statistics;
      // the PLINK loop skips it (:1870)
      if (!g.plink_nan_rule) {
        beta = NAN; se = NAN; p_wald = NAN;
      } else {
        wald_skipped = true;
      }
    } else {
      // CalcRLWald's CalcPab at lambda_remle is the evaluation LogRL_f already made there (:2104,:2122-2123)
      wald_score_from<M, false>(cx, at_remle, beta, se, p_wald);
    }
  }
  if (a_mode == 2 || a_mode == 4 || a_mode == 9) {
    Agg at_mle;
    calc_lambda<M, false>(cx, lambda_mle, logl_H1, at_mle);
    p_lrt = chisq_Q1_dev(2.0 * (logl_H1 - g.logl_mle_H0));
    if (isnan(logl_H1)) p_lrt = NAN;
  }
  if (g.plink_nan_rule && isnan(logl_H1)) p_wald = p_lrt = logl_H1; // src/lmm.cpp:1882-1884
  if (wald_skipped && a_mode == 1) {
    // AnalyzePlink keeps beta/se of the PREVIOUS SNP here (function-scope variables,
    // src/lmm.cpp:1725); plink_carry_kernel fills these two from the preceding SNP.
    beta = NAN;
    se = NAN;
  }
  if (lane == 0) {
    SumStat o;
    o.beta = beta; o.se = se; o.lambda_remle = lambda_remle; o.lambda_mle = lambda_mle;
    o.p_wald = p_wald; o.p_lrt = p_lrt; o.p_score = p_score; o.logl_H1 = logl_H1;
    g.out[snp] = o;
  }
}

template <int C>
__global__ __launch_bounds__(256, (C == 1 ? 3 : (C == 2 ? 2 : 1))) void lmm_assoc_kernel(AssocArgs g) {
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (snp >= g.l) return;
  assoc_one_snp(g, FixedC<C>(), snp, lane);
}
// tuning variants of the c = 1 kernel: streaming-loop unroll UNR, WPS wavefronts per SIMD (GEMMA_HIP_ASSOC_VARIANT)
template <int UNR, int WPS>
__global__ __launch_bounds__(256, WPS) void lmm_assoc1_variant_kernel(AssocArgs g) {
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (snp >= g.l) return;
  assoc_one_snp(g, FixedC<1, UNR>(), snp, lane);
}

__global__ __launch_bounds__(256) void lmm_assoc_generic_kernel(AssocArgs g, int c) {
  __shared__ double lds[4 * GEN_LDS_PER_WAVE];
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (snp >= g.l) return;
  GenericC m;
  m.cc = c;
  m.L = lds + (threadIdx.x >> 6) * GEN_LDS_PER_WAVE;
  assoc_one_snp(g, m, snp, lane);
}

// more than GEN_CMAX covariates: one wavefront per workgroup, tables in dynamic LDS (6 * gen_ni_for(c) doubles)
__global__ __launch_bounds__(64) void lmm_assoc_wide_kernel(AssocArgs g, int c) {
  extern __shared__ double wide_lds[];
  const long snp = blockIdx.x;
  if (snp >= g.l) return;
  GenericC m;
  m.cc = c;
  m.L = wide_lds;
  m.ni = gen_ni_for(c);
  assoc_one_snp(g, m, snp, threadIdx.x & 63);
}

// ------------------------------------------------------------------ AnalyzeGene (src/lmm.cpp:1365-1471)
// The roles are swapped: every row is a PHENOTYPE (a gene's expression over the analysed individuals, rotated:
// U^T y_g), the tested variable x is one fixed vector (g.Uty holds U^T x here).  Per row, as the reference:
//   a_mode 2/3/4/9: null ML search on (W, y_g)  -> l_H0, logl_H0   (:1424-1427; its FUNC_PARAM says calc_null = false
//                   with the x columns of Uab zero -- the x projection step is skipped (ps_ww == 0) and the ML formulas
//                   carry no df, so this IS the null model's ML fit; evaluated here as the c-1 covariate model + "x" = w_c)
//   a_mode 3/4/9:   CalcRLScore at l_H0 (the row's own null lambda, :1434-1436)
//   a_mode 1/4:     CalcLambda('R') + CalcRLWald;   a_mode 2/4/9: CalcLambda('L'), p_lrt against logl_H0
// MN: the model type with one covariate fewer (null fit).
template <class M, class MN>
__device__ __forceinline__ void gene_one_row(const AssocArgs &g, const M &model, const MN &null_model, long row, int lane) {
  const double *yrow = g.UtX + row * g.ld;
  const int a_mode = g.a_mode;
  double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
  double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0, l_H0 = 0.0, logl_H0 = 0.0;
  if (a_mode == 2 || a_mode == 3 || a_mode == 4 || a_mode == 9) {
    SnpCtx<MN> cn;
    cn.g = &g;
    cn.x = g.UtWt + (long)null_model.c() * g.n; // last covariate in the "x" slot
    cn.y = yrow;
    cn.lane = lane;
    cn.m = null_model;
    cn.logdet_iw = 0.0;
    cn.trow = nullptr;
    cn.cslots = nullptr;
    Agg tmp;
    calc_lambda<MN, false>(cn, l_H0, logl_H0, tmp);
  }
  SnpCtx<M> cx;
  cx.g = &g;
  cx.x = g.Uty; // the fixed tested variable
  cx.y = yrow;
  cx.lane = lane;
  cx.m = model;
  cx.logdet_iw = 0.0;
  cx.trow = nullptr;
  cx.cslots = nullptr;
  if (a_mode == 3 || a_mode == 4 || a_mode == 9) wald_score<M, true>(cx, l_H0, beta, se, p_score);
  if (a_mode == 1 || a_mode == 4) {
    cx.logdet_iw = logdet_iw_of(cx);
    Agg at_remle;
    calc_lambda<M, true>(cx, lambda_remle, logl_H1, at_remle);
    if (isnan(logl_H1)) { // CalcRLWald(NaN) (:1440)
      beta = NAN; se = NAN; p_wald = NAN;
    } else {
      wald_score_from<M, false>(cx, at_remle, beta, se, p_wald);
    }
  }
  if (a_mode == 2 || a_mode == 4 || a_mode == 9) {
    Agg at_mle;
    calc_lambda<M, false>(cx, lambda_mle, logl_H1, at_mle);
    p_lrt = chisq_Q1_dev(2.0 * (logl_H1 - logl_H0));
    if (isnan(logl_H1) || isnan(logl_H0)) p_lrt = NAN;
  }
  if (lane == 0) {
    SumStat o;
    o.beta = beta; o.se = se; o.lambda_remle = lambda_remle; o.lambda_mle = lambda_mle;
    o.p_wald = p_wald; o.p_lrt = p_lrt; o.p_score = p_score; o.logl_H1 = logl_H1;
    g.out[row] = o;
  }
}

template <int C>
__global__ __launch_bounds__(256, (C <= 2 ? 2 : 1)) void lmm_gene_kernel(AssocArgs g) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= g.l) return;
  gene_one_row(g, FixedC<C>(), FixedC<C - 1>(), row, lane);
}

__global__ __launch_bounds__(256) void lmm_gene_generic_kernel(AssocArgs g, int c) {
  __shared__ double lds[4 * GEN_LDS_PER_WAVE];
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= g.l) return;
  GenericC m, mn;
  m.cc = c;
  m.L = lds + (threadIdx.x >> 6) * GEN_LDS_PER_WAVE;
  mn.cc = c - 1;
  mn.L = m.L; // the two fits run one after the other
  gene_one_row(g, m, mn, row, lane);
}

__global__ __launch_bounds__(64) void lmm_gene_wide_kernel(AssocArgs g, int c) {
  extern __shared__ double wide_lds[];
  const long row = blockIdx.x;
  if (row >= g.l) return;
  GenericC m, mn;
  m.cc = c;
  m.L = wide_lds;
  m.ni = gen_ni_for(c);
  mn.cc = c - 1;
  mn.L = m.L; // the two fits run one after the other
  mn.ni = m.ni;
  gene_one_row(g, m, mn, row, threadIdx.x & 63);
}

// ------------------------------------------------------------------ GXE (src/lmm.cpp:2283-2608)
// AnalyzeBimbamGXE / AnalyzePlinkGXE: the covariates of SNP s are [W, env, x_s] (c + 2 of them) and the tested variable
// is x_s . env.  Here g.UtWt holds c + 1 shared rows (U^T W columns, then U^T env), g.UtX the rows U^T x_s (the LAST
// covariate), g.UtZ the rows U^T (x_s . env) (the x slot), g.flip the rows whose genotypes were recoded 2 - x
// (x_mean > 1, :2352-2354 / :2530-2532: beta changes sign, :2403 / :2584).  Per SNP, as the reference:
//   a_mode 2/4:   CalcLambda('L') on the c + 2 covariates alone (calc_null = true, :2384-2387) -> logl_H0
//                 (a_mode 9 does NOT compute it: logl_H0 stays 0 there, as in the reference)
//   a_mode 3/4/9: CalcRLScore at the GLOBAL l_mle_null (:2396)
//   a_mode 1/4:   CalcLambda('R') + CalcRLWald;  a_mode 2/4/9: CalcLambda('L'), p_lrt against logl_H0
// M: c + 2 covariates with the per-SNP last one; MN: c + 1 shared covariates with x_s in the x slot (the null fit).
template <class M, class MN>
__device__ __forceinline__ void gxe_one_snp(const AssocArgs &g, M model, const MN &null_model, long snp, int lane) {
  const double *xrow = g.UtX + snp * g.ld, *zrow = g.UtZ + snp * g.ld;
  const int a_mode = g.a_mode;
  double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
  double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0, logl_H0 = 0.0;
  if (a_mode == 2 || a_mode == 4) {
    SnpCtx<MN> cn;
    cn.g = &g;
    cn.x = xrow; // the SNP is the last covariate of the null model
    cn.y = g.Uty;
    cn.lane = lane;
    cn.m = null_model;
    cn.logdet_iw = 0.0;
    cn.trow = nullptr;
    cn.cslots = nullptr;
    Agg tmp;
    calc_lambda<MN, false>(cn, lambda_mle, logl_H0, tmp);
  }
  model.wlast = xrow;
  SnpCtx<M> cx;
  cx.g = &g;
  cx.x = zrow;
  cx.y = g.Uty;
  cx.lane = lane;
  cx.m = model;
  cx.logdet_iw = 0.0;
  cx.trow = nullptr;
  cx.cslots = nullptr;
  if (a_mode == 3 || a_mode == 4 || a_mode == 9) wald_score<M, true>(cx, g.l_mle_null, beta, se, p_score);
  if (a_mode == 1 || a_mode == 4) {
    cx.logdet_iw = logdet_iw_of(cx);
    Agg at_remle;
    calc_lambda<M, true>(cx, lambda_remle, logl_H1, at_remle);
    if (isnan(logl_H1)) {
      beta = NAN; se = NAN; p_wald = NAN;
    } else {
      wald_score_from<M, false>(cx, at_remle, beta, se, p_wald);
    }
  }
  if (a_mode == 2 || a_mode == 4 || a_mode == 9) {
    Agg at_mle;
    calc_lambda<M, false>(cx, lambda_mle, logl_H1, at_mle);
    p_lrt = chisq_Q1_dev(2.0 * (logl_H1 - logl_H0));
    if (isnan(logl_H1) || isnan(logl_H0)) p_lrt = NAN;
  }
  if (g.flip[snp]) beta *= -1;
  if (lane == 0) {
    SumStat o;
    o.beta = beta; o.se = se; o.lambda_remle = lambda_remle; o.lambda_mle = lambda_mle;
    o.p_wald = p_wald; o.p_lrt = p_lrt; o.p_score = p_score; o.logl_H1 = logl_H1;
    g.out[snp] = o;
  }
}

// CT = c + 2 (covariates of the alternative model), register kernel for CT <= 4
template <int CT>
__global__ __launch_bounds__(256, (CT <= 3 ? 2 : 1)) void lmm_gxe_kernel(AssocArgs g) {
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (snp >= g.l) return;
  gxe_one_snp(g, FixedC<CT, 2, true>(), FixedC<CT - 1>(), snp, lane);
}

__global__ __launch_bounds__(256) void lmm_gxe_generic_kernel(AssocArgs g, int ct) {
  __shared__ double lds[4 * GEN_LDS_PER_WAVE];
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (snp >= g.l) return;
  GenericC m, mn;
  m.cc = ct;
  m.L = lds + (threadIdx.x >> 6) * GEN_LDS_PER_WAVE;
  mn.cc = ct - 1;
  mn.L = m.L;
  gxe_one_snp(g, m, mn, snp, lane);
}

// sum_i log|l*delta_i + 1| at l_min and l_max, accumulated exactly like the LOGDET branch of the row passes
// (lane-strided partial sums, 64-lane butterfly) so that the constants equal what a SNP's own pass would give
__global__ __launch_bounds__(64) void logdet_ends_kernel(const double *__restrict__ eval, int n, double l_min,
                                                         double l_max, double *__restrict__ out2) {
  const int lane = threadIdx.x & 63;
  double a = 0.0, b = 0.0;
  for (int i = lane; i < n; i += 64) {
    const double d = eval[i];
    a += log(fabs(d * l_min + 1.0));
    b += log(fabs(d * l_max + 1.0));
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (lane == 0) {
    out2[0] = a;
    out2[1] = b;
  }
}

// ------------------------------------------------------------------ null model
// CalcLambda(func, eval, UtW, Uty, ...) with calc_null = true (GEMMA src/lmm.cpp:2143-2180,
// called at src/gemma.cpp:2711,2734), CalcPve's LogRL_dev2 (:2197) and CalcLmmVgVeBeta's P_yy
// (:2253-2258).  Projecting out w_1..w_c (nc_total = c, df = n - c) is the alternative-model code
// with c - 1 covariates and the last covariate in the role of x.
struct NullOut {
  double l_mle, logl_mle, l_remle, logl_remle, dev2_remle, Pyy_remle, Pyy_mle;
};

template <class M>
__device__ __forceinline__ void null_model(const AssocArgs &g, const M &model, int cp, int lane, NullOut *out) {
  SnpCtx<M> cx;
  cx.g = &g;
  cx.x = g.UtWt + (long)cp * g.n; // last covariate column
  cx.y = g.Uty;
  cx.lane = lane;
  cx.m = model;
  cx.trow = nullptr;
  cx.cslots = nullptr;
  cx.logdet_iw = logdet_iw_of(cx);
  NullOut o;
  Agg tmp;
  calc_lambda<M, false>(cx, o.l_mle, o.logl_mle, tmp);
  calc_lambda<M, true>(cx, o.l_remle, o.logl_remle, tmp);
  double d1, d2;
  deriv<M, true, 3>(cx, o.l_remle, d1, d2);
  o.dev2_remle = d2;
  Agg A;
  cx.m.template eval<1, false>(g, cx.x, cx.y, o.l_remle, lane, A);
  o.Pyy_remle = uniform(A.yy1);
  cx.m.template eval<1, false>(g, cx.x, cx.y, o.l_mle, lane, A);
  o.Pyy_mle = uniform(A.yy1);
  if (lane == 0) *out = o;
}

__global__ __launch_bounds__(64) void lmm_gxe_wide_kernel(AssocArgs g, int ct) {
  extern __shared__ double wide_lds[];
  const long snp = blockIdx.x;
  if (snp >= g.l) return;
  GenericC m, mn;
  m.cc = ct;
  m.L = wide_lds;
  m.ni = gen_ni_for(ct);
  mn.cc = ct - 1;
  mn.L = m.L;
  mn.ni = m.ni;
  gxe_one_snp(g, m, mn, snp, threadIdx.x & 63);
}
__global__ __launch_bounds__(64) void lmm_null_wide_kernel(AssocArgs g, int cp, NullOut *out) {
  extern __shared__ double wide_lds[];
  GenericC m;
  m.cc = cp;
  m.L = wide_lds;
  m.ni = gen_ni_for(cp);
  null_model(g, m, cp, threadIdx.x & 63, out);
}
template <int CP>
__global__ __launch_bounds__(64) void lmm_null_kernel(AssocArgs g, NullOut *out) {
  null_model(g, FixedC<CP>(), CP, threadIdx.x & 63, out);
}

__global__ __launch_bounds__(64) void lmm_null_generic_kernel(AssocArgs g, int cp, NullOut *out) {
  __shared__ double lds[GEN_LDS_PER_WAVE];
  GenericC m;
  m.cc = cp;
  m.L = lds;
  null_model(g, m, cp, threadIdx.x & 63, out);
}

} // namespace gemma_hip
