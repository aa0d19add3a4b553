// Multivariate LMM, per-SNP stage (GEMMA src/mvlmm.cpp:3287-3374): MphEM (:599-724), MphCalcP (:727-831)
// and MphNR (:2608-2760) for one SNP per WAVEFRONT.
//
// Everything is done in the basis that diagonalises all H_k = delta_k V_g + V_e at once (EigenProc :213-282):
// H_k^-1 = UltVehi^T diag(1 / (delta_k D_l + 1)) UltVehi.  There the dc x dc matrix Q of CalcQi (:285-329) is a
// direct sum of d blocks of size c x c (its entries couple (covariate i, component l) with (covariate j, component l)
// only), so "LU-invert Q" becomes d small SPD inversions, the REML part of CalcSigma (:517-548) and the x P x of
// MphCalcP are diagonal, and the whole EM iteration is two passes (REML; three for ML) over the n individuals with
// < 50 running sums, which the 64 lanes of a wavefront stride over and butterfly-reduce.  In the fixed form (template extents
// DT, CT > 0) the small matrices live in registers and every loop over them unrolls; in the run-time form (DT = CT = 0, see
// below) they live in private memory.  All lanes carry the same copy.
//
// The lane policy makes the same source run on one CPU "lane" in tests/host/mvlmm_harness.cpp (test infrastructure:
// the shipped library has no CPU path).
#pragma once
#include <math.h>

#ifndef MV_HD
#define MV_HD __device__ __forceinline__
#endif
// the run-time instance calls its large pieces instead of inlining them: one copy of each, and private-memory frames that overlap
// (inlined, the 21 call sites of one SNP summed to 120 KB of scratch per lane and 8 minutes of compile time)
#ifndef MV_OUTLINE
#define MV_OUTLINE __device__ __noinline__
#endif

namespace gemma_hip {

// Every function below exists in two forms from one source: FIXED (template arguments DT phenotypes, CT rows of X = covariates + the
// SNP: loops unroll, the small matrices live in registers -- the kernels of mvlmm_kernels*.hip) and RUN-TIME (DT = CT = 0: d and c
// come from MvRt, arrays are sized by the caps below and indexed with the run-time strides -- one kernel for every other (d, c),
// and the only one that takes a second SNP row, the gene-environment interaction of MVLMM::AnalyzeBimbamGXE, src/mvlmm.cpp:3970).
constexpr int MV_DMAX = 8;   // phenotypes
constexpr int MV_CMAX = 12;  // rows of X: covariates (+ the environment) + the SNP (+ its interaction row)
constexpr int MV_BMAX = MV_DMAX * MV_CMAX; // entries of B
template <int DT> constexpr int mv_dk = DT > 0 ? DT : MV_DMAX; // array extents of the two forms
template <int CT> constexpr int mv_ck = CT > 0 ? CT : MV_CMAX;
template <int MT> constexpr int mv_mk = MT > 0 ? MT : -MT;     // small dense algebra: MT > 0 fixed order, MT < 0 run-time order <= -MT
template <int XT, int CAP> constexpr int mv_mt = XT > 0 ? XT : -CAP;

// run-time shape (read by the DT = CT = 0 instance only) and the second SNP row of the interaction test
struct MvRt {
  int d = 0, c = 0;
  const double *x2 = nullptr; // GXE: row c - 2 of X is the SNP (the `x` argument), row c - 1 this one (U^T (x o env))
};
#define MV_SHAPE(rt_)                                                                                                      \
  [[maybe_unused]] constexpr int DK = mv_dk<DT>, CK = mv_ck<CT>, TK = CK * (CK + 1) / 2, TDK = DK * (DK + 1) / 2, VSK = TDK, \
                                 H2K = 2 * VSK;                                                                            \
  [[maybe_unused]] const int D = DT > 0 ? DT : (rt_).d, C = CT > 0 ? CT : (rt_).c, T = C * (C + 1) / 2,                     \
                             TD = D * (D + 1) / 2, VS = TD, H2 = 2 * VS
#define MV_DD (mv_dk<DT> * mv_dk<DT>)
#define MV_DC (mv_dk<DT> * mv_ck<CT>)
#define MV_CC (mv_ck<CT> * mv_ck<CT>)
#define MV_D1 (mv_dk<DT>)

struct MvArgs {
  const double *UtX;  // l x ld, SNP-major
  long ld, l;
  int n;
  const double *eval;  // n
  const double *Wt;    // (c - 1) x n : U^T W transposed
  const double *Yt;    // d x n       : U^T Y transposed
  double Vg_null[MV_DMAX * MV_DMAX], Ve_null[MV_DMAX * MV_DMAX], B_null[MV_BMAX]; // B_null: d x (c - 1)
  double logl_H0;      // MLE null log-likelihood (the LRT reference, :3317)
  int a_mode;
  int em_iter;         // per-SNP cap (the caller passes em_iter / 10, :3310)
  double em_prec;      // em_prec * 10
  int nr_iter;         // nr_iter / 10
  double nr_prec;      // nr_prec * 10
  double p_nr;
  double *out;         // l x stride: beta[d], Vbeta[v], Vg[v], Ve[v], p_wald, p_lrt, p_score
  int stride;
  int crt;             // -crt: Edgeworth-corrected p values (CalcCRT / PCRT) for the SNPs that reach the Newton-Raphson stage
  // run-time instance only
  int d = 0, c = 0;    // phenotypes; rows of X of the alternative model (covariates + SNP [+ interaction])
  const double *UtX2 = nullptr; // GXE: U^T (x o env), l x ld like UtX; Wt then ends with the U^T env row
  const int *flip = nullptr;    // GXE: SNPs whose allele was switched (x_mean > 1: x <- 2 - x, :4232-4236): beta changes sign (:4331-4333)
  double *scratch = nullptr;    // MvNrLayout(d, c).DOUBLES doubles per workgroup (the Newton-Raphson tables: too large for LDS at the caps)
};

#ifdef __HIPCC__
struct MvWaveLanes {
  static constexpr int N = 64;
  static MV_HD int lane() { return (int)(threadIdx.x & 63); }
  static MV_HD double sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  }
};
#endif

// ---------------------------------------------------------------- small dense algebra (registers)
// cyclic Jacobi, A = V diag(w) V^T, eigenvalues ascending (LAPACK's order), each vector signed so that its entry of
// largest magnitude is positive.  The ML EM subtracts the PREVIOUS iteration's U_l^T V_e^-1/2 B X from this
// iteration's rotated phenotypes (UltVehiBX is not refreshed before UpdateU, src/mvlmm.cpp:679-686), so the component
// order and signs must not jump between two nearly equal matrices.
template <int MT>
MV_HD void mv_jacobi(const double (&A)[mv_mk<MT> * mv_mk<MT>], double (&w)[mv_mk<MT>], double (&V)[mv_mk<MT> * mv_mk<MT>], int m) {
  constexpr int MK = mv_mk<MT>;
  const int M = MT > 0 ? MT : m;
  double a[MK * MK];
#pragma unroll
  for (int i = 0; i < M * M; ++i) a[i] = A[i];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) V[i * M + j] = (i == j) ? 1.0 : 0.0;
  if (M > 1) {
    for (int sweep = 0; sweep < 60; ++sweep) {
      double off = 0.0, diag = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        diag += a[i * M + i] * a[i * M + i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) off += a[i * M + j] * a[i * M + j];
      }
      if (off <= 1e-34 * diag || off == 0.0) break;
#pragma unroll
      for (int p = 0; p < M; ++p)
#pragma unroll
        for (int q = p + 1; q < M; ++q) {
          const double apq = a[p * M + q];
          if (apq != 0.0) {
            const double theta = (a[q * M + q] - a[p * M + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
            for (int k = 0; k < M; ++k) {
              const double akp = a[k * M + p], akq = a[k * M + q];
              a[k * M + p] = cs * akp - sn * akq;
              a[k * M + q] = sn * akp + cs * akq;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) {
              const double apk = a[p * M + k], aqk = a[q * M + k];
              a[p * M + k] = cs * apk - sn * aqk;
              a[q * M + k] = sn * apk + cs * aqk;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) {
              const double vkp = V[k * M + p], vkq = V[k * M + q];
              V[k * M + p] = cs * vkp - sn * vkq;
              V[k * M + q] = sn * vkp + cs * vkq;
            }
          }
        }
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i) w[i] = a[i * M + i];
#pragma unroll
  for (int i = 0; i < M; ++i) // selection sort, ascending; columns follow
#pragma unroll
    for (int j = i + 1; j < M; ++j)
      if (w[j] < w[i]) {
        double t = w[i];
        w[i] = w[j];
        w[j] = t;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          t = V[k * M + i];
          V[k * M + i] = V[k * M + j];
          V[k * M + j] = t;
        }
      }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    double big = 0.0, sgn = 1.0;
#pragma unroll
    for (int k = 0; k < M; ++k)
      if (fabs(V[k * M + i]) > big) {
        big = fabs(V[k * M + i]);
        sgn = V[k * M + i] < 0 ? -1.0 : 1.0;
      }
#pragma unroll
    for (int k = 0; k < M; ++k) V[k * M + i] *= sgn;
  }
}

// inverse and log-determinant of a symmetric positive definite M x M matrix (Cholesky); a non-positive pivot
// propagates NaN like the reference's LU of a singular Q propagates inf/NaN
template <int MT> MV_HD double mv_spd_inverse(const double (&A)[mv_mk<MT> * mv_mk<MT>], double (&Ai)[mv_mk<MT> * mv_mk<MT>], int m) {
  constexpr int MK = mv_mk<MT>;
  const int M = MT > 0 ? MT : m;
  double L[MK * MK], Li[MK * MK];
  double lndet = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    double s = A[j * M + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * M + k] * L[j * M + k];
    const double ljj = sqrt(s);
    L[j * M + j] = ljj;
    lndet += 2.0 * log(ljj);
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      double t = A[i * M + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * M + k] * L[j * M + k];
      L[i * M + j] = t / ljj;
    }
  }
#pragma unroll
  for (int j = 0; j < M; ++j) { // Li = L^-1 (lower)
    Li[j * M + j] = 1.0 / L[j * M + j];
#pragma unroll
    for (int i = j + 1; i < M; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) t -= L[i * M + k] * Li[k * M + j];
      Li[i * M + j] = t / L[i * M + i];
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double t = 0.0;
#pragma unroll
      for (int k = i; k < M; ++k) t += Li[k * M + i] * Li[k * M + j];
      Ai[i * M + j] = t;
      Ai[j * M + i] = t;
    }
  return lndet;
}

// gsl_cdf_chisq_Q(x, nu), integer nu: finite sums for integer / half-integer shape (all terms positive)
MV_HD double mv_chisq_Q(double x, int nu) {
  if (!(x > 0.0)) return (x == x) ? 1.0 : x;
  const double y = 0.5 * x;
  if ((nu & 1) == 0) {
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < nu / 2; ++k) {
      term *= y / k;
      sum += term;
    }
    return exp(-y) * sum;
  }
  double sum = 0.0, term = sqrt(y) / 0.886226925452758013649;
  for (int k = 1; k <= (nu - 1) / 2; ++k) {
    sum += term;
    term *= y / (k + 0.5);
  }
  return erfc(sqrt(y)) + exp(-y) * sum;
}

// gsl_cdf_ugaussian_Qinv to ~1e-9 (Acklam's rational approximation) + one Newton step on erfc: only the start value of the
// iteration below
MV_HD double mv_ugaussian_Qinv(double Q) {
  const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                       1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
  const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                       6.680131188771972e+01, -1.328068155288572e+01};
  const double c[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                       -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
  const double d[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
  const double p = 1.0 - Q;
  double x;
  if (p < 0.02425) {
    const double q = sqrt(-2.0 * log(p));
    x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
  } else if (p <= 1.0 - 0.02425) {
    const double q = p - 0.5, r = q * q;
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
  } else {
    const double q = sqrt(-2.0 * log(1.0 - p));
    x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
  }
  const double e = 0.5 * erfc(x * 0.70710678118654752440) - Q;
  return x + e / (exp(-0.5 * x * x) * 0.39894228040143267794);
}

// gsl_cdf_chisq_Qinv(Q, nu) = 2 gsl_cdf_gamma_Qinv(Q, nu / 2, 1) (GSL cdf/gammainv.c): start value by range of Q, then its
// Lagrange-corrected Newton step until |step| <= 1e-10 x (at most 33 rounds)
MV_HD double mv_chisq_Qinv(double Q, int nu) {
  const double a = 0.5 * (double)nu;
  if (Q == 1.0) return 0.0;
  if (Q == 0.0) return 1.0 / 0.0;
  const double lg = lgamma(a);
  double x;
  if (Q < 0.05) x = -log(Q) + lg;
  else if (Q > 0.95) x = exp((lg + log1p(-Q)) / a);
  else {
    const double xg = mv_ugaussian_Qinv(Q);
    x = (xg < -0.5 * sqrt(a)) ? a : sqrt(a) * xg + a;
  }
  for (int n = 0;;) {
    const double dQ = Q - mv_chisq_Q(2.0 * x, nu);
    const double phi = exp((a - 1.0) * log(x) - x - lg);
    if (dQ == 0.0 || n++ > 32) break;
    const double lambda = -dQ / fmax(2.0 * fabs(dQ / x), phi);
    const double step0 = lambda, step1 = -((a - 1.0) / x - 1.0) * lambda * lambda / 4.0;
    double step = step0;
    if (fabs(step1) < 0.5 * fabs(step0)) step += step1;
    if (x + step > 0) x += step;
    else x /= 2.0;
    if (!(fabs(step0) > 1e-10 * x)) break;
  }
  return 2.0 * x;
}

// PCRT (src/mvlmm.cpp:2952-2970): mode 1 Wald, 2 LRT, 3 score
MV_HD double mv_pcrt(int mode, int d, double p_value, const double (&crt)[3]) {
  const double q = (double)d;
  const double chisq = mv_chisq_Qinv(p_value, d);
  double chisq_crt;
  if (mode == 1) {
    const double a = crt[2] / (2.0 * q * (q + 2.0)), b = 1.0 + (crt[0] + crt[1]) / (2.0 * q);
    const double rad = b * b + 4.0 * a * chisq;
    chisq_crt = (-1.0 * b + (rad >= 0.0 ? sqrt(rad) : (0.0 / 0.0))) / (2.0 * a); // safe_sqrt: NaN below zero
  } else if (mode == 2) {
    chisq_crt = chisq / (1.0 + crt[0] / (2.0 * q));
  } else {
    chisq_crt = chisq;
  }
  return mv_chisq_Q(chisq_crt, d);
}

// ---------------------------------------------------------------- per-SNP state
template <int DT> struct MvBasis { // EigenProc, src/mvlmm.cpp:213-282
  static constexpr int DK = mv_dk<DT>;
  double Dl[DK], UltVeh[DK * DK], UltVehi[DK * DK], logdet_Ve;
  MV_HD void build(const double (&Vg)[MV_DD], const double (&Ve)[MV_DD], int d) {
    const int D = DT > 0 ? DT : d;
    double w[DK], Ul[DK * DK], Veh[DK * DK], Vehi[DK * DK], T1[DK * DK], Lam[DK * DK];
    mv_jacobi<mv_mt<DT, MV_DMAX>>(Ve, w, Ul, D);
    logdet_Ve = 0.0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) Veh[i] = Vehi[i] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (w[i] > 0) {
        logdet_Ve += log(w[i]);
        const double s = sqrt(w[i]), si = 1.0 / s;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) {
            Veh[a * D + b] += s * Ul[a * D + i] * Ul[b * D + i];
            Vehi[a * D + b] += si * Ul[a * D + i] * Ul[b * D + i];
          }
      }
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) s += Vg[a * D + t] * Vehi[t * D + b];
        T1[a * D + b] = s;
      }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) s += Vehi[a * D + t] * T1[t * D + b];
        Lam[a * D + b] = s;
      }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a + 1; b < D; ++b) Lam[a * D + b] = Lam[b * D + a] = 0.5 * (Lam[a * D + b] + Lam[b * D + a]);
    mv_jacobi<mv_mt<DT, MV_DMAX>>(Lam, Dl, Ul, D);
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (Dl[i] < 0) Dl[i] = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) {
          s1 += Ul[t * D + a] * Veh[t * D + b];
          s2 += Ul[t * D + a] * Vehi[t * D + b];
        }
        UltVeh[a * D + b] = s1;
        UltVehi[a * D + b] = s2;
      }
  }
};

// the sums of one pass over the individuals at a fixed basis: per component l the c x c block Q_l (upper triangle),
// xHiy (c per component) and the scalar part of MphCalcLogL (:571-581)
template <int DT, int CT> struct MvMoments {
  static constexpr int DK = mv_dk<DT>, CK = mv_ck<CT>, TK = CK * (CK + 1) / 2;
  double Q[DK * TK], xHiy[DK * CK], ll;
};

// row k of X^T: the covariates from Wt, then the SNP (fixed form: always one row; run-time form with rt.x2: the SNP and its
// interaction row)
template <int CT>
MV_HD void mv_load_x(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, int k, double (&xv)[mv_ck<CT>]) {
  const int C = CT > 0 ? CT : rt.c;
  const bool two = CT == 0 && rt.x2 != nullptr;
  const int nw = two ? C - 2 : C - 1;
#pragma unroll
  for (int j = 0; j < nw; ++j) xv[j] = g.Wt[(long)j * g.n + k];
  if (two) {
    xv[C - 2] = x[k];
    xv[C - 1] = rt.x2[k];
  } else {
    xv[C - 1] = x[k];
  }
}

template <int DT, int CT, class Lanes>
MV_HD void mv_pass_moments(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, const MvBasis<DT> &bs, MvMoments<DT, CT> &m) {
  MV_SHAPE(rt);
#pragma unroll
  for (int i = 0; i < D * T; ++i) m.Q[i] = 0.0;
#pragma unroll
  for (int i = 0; i < D * C; ++i) m.xHiy[i] = 0.0;
  m.ll = 0.0;
  for (int k = Lanes::lane(); k < g.n; k += Lanes::N) {
    const double delta = g.eval[k];
    double xv[CK], yv[DK];
    mv_load_x<CT>(g, rt, x, k, xv);
#pragma unroll
    for (int i = 0; i < D; ++i) yv[i] = g.Yt[(long)i * g.n + k];
#pragma unroll
    for (int l = 0; l < D; ++l) {
      const double dd = delta * bs.Dl[l] + 1.0, w = 1.0 / dd;
      double yt = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) yt += bs.UltVehi[l * D + i] * yv[i];
      m.ll += yt * yt * w + log(dd);
      int t = 0;
#pragma unroll
      for (int a = 0; a < C; ++a) {
        const double xa = xv[a] * w;
        m.xHiy[l * C + a] += xa * yt;
#pragma unroll
        for (int b = a; b < C; ++b, ++t) m.Q[l * T + t] += xa * xv[b];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < D * T; ++i) m.Q[i] = Lanes::sum(m.Q[i]);
#pragma unroll
  for (int i = 0; i < D * C; ++i) m.xHiy[i] = Lanes::sum(m.xHiy[i]);
  m.ll = Lanes::sum(m.ll);
}

template <int CT> MV_HD void mv_unpack_sym(const double *tri, double (&A)[MV_CC], int c) {
  const int C = CT > 0 ? CT : c;
  int t = 0;
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = a; b < C; ++b, ++t) A[a * C + b] = A[b * C + a] = tri[t];
}

// log|X X^T| and (X X^T)^-1 over the c covariate rows (:631-649)
template <int CT, class Lanes>
MV_HD double mv_xxt(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, double (&XXti)[MV_CC]) {
  constexpr int CK = mv_ck<CT>, TK = CK * (CK + 1) / 2;
  const int C = CT > 0 ? CT : rt.c, T = C * (C + 1) / 2;
  double tri[TK];
#pragma unroll
  for (int i = 0; i < T; ++i) tri[i] = 0.0;
  for (int k = Lanes::lane(); k < g.n; k += Lanes::N) {
    double xv[CK];
    mv_load_x<CT>(g, rt, x, k, xv);
    int t = 0;
#pragma unroll
    for (int a = 0; a < C; ++a)
#pragma unroll
      for (int b = a; b < C; ++b, ++t) tri[t] += xv[a] * xv[b];
  }
#pragma unroll
  for (int i = 0; i < T; ++i) tri[i] = Lanes::sum(tri[i]);
  double XXt[CK * CK];
  mv_unpack_sym<CT>(tri, XXt, C);
  return mv_spd_inverse<mv_mt<CT, MV_CMAX>>(XXt, XXti, C);
}

// MphEM, src/mvlmm.cpp:599-724.  Vg, Ve (d x d) and B (d x c) are updated in place; returns the last logl.
template <int DT, int CT, class Lanes>
MV_HD double mv_em(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, bool reml, int max_iter, double max_prec,
                   double lndet_xxt, const double (&XXti)[MV_CC], double (&Vg)[MV_DD], double (&Ve)[MV_DD],
                   double (&B)[MV_DC]) {
  MV_SHAPE(rt);
  constexpr double LOG2PI = 1.8378770664093454836;
  const int n = g.n;
  const double logl_const = reml ? -0.5 * (double)(n - C) * (double)D * LOG2PI + 0.5 * (double)D * lndet_xxt
                                 : -0.5 * (double)n * (double)D * LOG2PI;
  double UltVehiB[DK * CK]; // component-major: [l][j]
#pragma unroll
  for (int i = 0; i < D * C; ++i) UltVehiB[i] = 0.0;
  double logl_old = 0.0, logl_new = 0.0;
  MvBasis<DT> bs;
  MvMoments<DT, CT> m;
  double Qi[DK][CK * CK];
  for (int t = 0; t < max_iter; ++t) {
    bs.build(Vg, Ve, D);
    mv_pass_moments<DT, CT, Lanes>(g, rt, x, bs, m);
    double logdet_Q = 0.0, quad = 0.0, bl[DK * CK];
#pragma unroll
    for (int l = 0; l < D; ++l) {
      double Ql[CK * CK];
      mv_unpack_sym<CT>(m.Q + l * T, Ql, C);
      logdet_Q += mv_spd_inverse<mv_mt<CT, MV_CMAX>>(Ql, Qi[l], C);
#pragma unroll
      for (int a = 0; a < C; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < C; ++b) s += Qi[l][a * C + b] * m.xHiy[l * C + b];
        bl[l * C + a] = s;
        quad += s * m.xHiy[l * C + a];
      }
    }
    logl_new = logl_const - 0.5 * (m.ll - quad) - 0.5 * (double)n * bs.logdet_Ve;
    if (reml) logl_new += -0.5 * (logdet_Q - (double)C * bs.logdet_Ve);
    if (t != 0 && fabs(logl_new - logl_old) < max_prec) break;
    logl_old = logl_new;
    // UltVehiB used by UpdateU (:674-686)
    if (reml) {
#pragma unroll
      for (int i = 0; i < D * C; ++i) UltVehiB[i] = bl[i];
    } else if (t == 0) {
#pragma unroll
      for (int l = 0; l < D; ++l)
#pragma unroll
        for (int j = 0; j < C; ++j) {
          double s = 0.0;
#pragma unroll
          for (int i = 0; i < D; ++i) s += bs.UltVehi[l * D + i] * B[i * C + j];
          UltVehiB[l * C + j] = s;
        }
    }
    double Bnew[DK * CK]; // UltVehiB after UpdateL_B (ML) -- equal to UltVehiB for REML
#pragma unroll
    for (int i = 0; i < D * C; ++i) Bnew[i] = UltVehiB[i];
    if (!reml) { // UpdateL_B :402-418: (UltVehiY - UltVehiU) X^T (X X^T)^-1
      double YUX[DK * CK];
#pragma unroll
      for (int i = 0; i < D * C; ++i) YUX[i] = 0.0;
      for (int k = Lanes::lane(); k < n; k += Lanes::N) {
        const double delta = g.eval[k];
        double xv[CK], yv[DK];
        mv_load_x<CT>(g, rt, x, k, xv);
#pragma unroll
        for (int i = 0; i < D; ++i) yv[i] = g.Yt[(long)i * n + k];
#pragma unroll
        for (int l = 0; l < D; ++l) {
          double yt = 0.0, bx = 0.0;
#pragma unroll
          for (int i = 0; i < D; ++i) yt += bs.UltVehi[l * D + i] * yv[i];
#pragma unroll
          for (int j = 0; j < C; ++j) bx += UltVehiB[l * C + j] * xv[j];
          const double oe = delta * bs.Dl[l] / (delta * bs.Dl[l] + 1.0);
          const double r = yt - (yt - bx) * oe;
#pragma unroll
          for (int j = 0; j < C; ++j) YUX[l * C + j] += r * xv[j];
        }
      }
#pragma unroll
      for (int i = 0; i < D * C; ++i) YUX[i] = Lanes::sum(YUX[i]);
#pragma unroll
      for (int l = 0; l < D; ++l)
#pragma unroll
        for (int j = 0; j < C; ++j) {
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < C; ++a) s += YUX[l * C + a] * XXti[a * C + j];
          Bnew[l * C + j] = s;
        }
    }
    // U_hat, E_hat, Sigma (UpdateU / UpdateE / CalcSigma / UpdateV :686-708)
    double VgS[TDK], VeS[TDK], Suu[DK], See[DK];
#pragma unroll
    for (int i = 0; i < TD; ++i) VgS[i] = VeS[i] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) Suu[i] = See[i] = 0.0;
    for (int k = Lanes::lane(); k < n; k += Lanes::N) {
      const double delta = g.eval[k];
      double xv[CK], yv[DK], Uv[DK], Ev[DK];
      mv_load_x<CT>(g, rt, x, k, xv);
#pragma unroll
      for (int i = 0; i < D; ++i) yv[i] = g.Yt[(long)i * n + k];
#pragma unroll
      for (int l = 0; l < D; ++l) {
        const double w = 1.0 / (delta * bs.Dl[l] + 1.0), ou = bs.Dl[l] * w, oe = delta * ou;
        double yt = 0.0, bx0 = 0.0, bx1 = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) yt += bs.UltVehi[l * D + i] * yv[i];
#pragma unroll
        for (int j = 0; j < C; ++j) {
          bx0 += UltVehiB[l * C + j] * xv[j];
          bx1 += Bnew[l * C + j] * xv[j];
        }
        Uv[l] = (yt - bx0) * oe;
        Ev[l] = yt - bx1 - Uv[l];
        double su = ou, se = oe;
        if (reml) { // x^T Q_l^-1 x
          double q = 0.0;
#pragma unroll
          for (int a = 0; a < C; ++a)
#pragma unroll
            for (int b = 0; b < C; ++b) q += xv[a] * Qi[l][a * C + b] * xv[b];
          su += delta * ou * ou * q;
          se += w * w * q;
        }
        Suu[l] += su;
        See[l] += se;
      }
      double Uh[DK], Eh[DK];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) {
          s1 += bs.UltVeh[l * D + a] * Uv[l];
          s2 += bs.UltVeh[l * D + a] * Ev[l];
        }
        Uh[a] = s1;
        Eh[a] = s2;
      }
      const double di = (delta != 0.0) ? 1.0 / delta : 0.0;
      int t2 = 0;
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b, ++t2) {
          VgS[t2] += Uh[a] * Uh[b] * di;
          VeS[t2] += Eh[a] * Eh[b];
        }
    }
#pragma unroll
    for (int i = 0; i < TD; ++i) {
      VgS[i] = Lanes::sum(VgS[i]);
      VeS[i] = Lanes::sum(VeS[i]);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      Suu[i] = Lanes::sum(Suu[i]);
      See[i] = Lanes::sum(See[i]);
    }
#pragma unroll
    for (int i = 0; i < D * C; ++i) UltVehiB[i] = Bnew[i];
    // B = UltVeh^T UltVehiB; V = (sums + UltVeh^T diag(S) UltVeh) / n
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) s += bs.UltVeh[l * D + a] * UltVehiB[l * C + j];
        B[a * C + j] = s;
      }
    int t2 = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b, ++t2) {
        double su = 0.0, se = 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) {
          su += bs.UltVeh[l * D + a] * Suu[l] * bs.UltVeh[l * D + b];
          se += bs.UltVeh[l * D + a] * See[l] * bs.UltVeh[l * D + b];
        }
        Vg[a * D + b] = Vg[b * D + a] = (VgS[t2] + su) / (double)n;
        Ve[a * D + b] = Ve[b * D + a] = (VeS[t2] + se) / (double)n;
      }
  }
  return logl_new;
}

template <int DT, int CT, class Lanes>
MV_OUTLINE double mv_em_out(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, bool reml, int max_iter, double max_prec,
                            double lndet_xxt, const double (&XXti)[MV_CC], double (&Vg)[MV_DD], double (&Ve)[MV_DD], double (&B)[MV_DC]) {
  return mv_em<DT, CT, Lanes>(g, rt, x, reml, max_iter, max_prec, lndet_xxt, XXti, Vg, Ve, B);
}
template <int DT, int CT, class Lanes>
MV_HD double mv_em_sel(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, bool reml, int max_iter, double max_prec,
                       double lndet_xxt, const double (&XXti)[MV_CC], double (&Vg)[MV_DD], double (&Ve)[MV_DD], double (&B)[MV_DC]) {
  if constexpr (DT == 0) return mv_em_out<0, 0, Lanes>(g, rt, x, reml, max_iter, max_prec, lndet_xxt, XXti, Vg, Ve, B);
  else return mv_em<DT, CT, Lanes>(g, rt, x, reml, max_iter, max_prec, lndet_xxt, XXti, Vg, Ve, B);
}

// MphCalcP, src/mvlmm.cpp:727-831: the covariates are the first C - 1 rows, the SNP the last; beta (d), Vbeta (d x d)
template <int DT, int CT, class Lanes>
MV_HD double mv_calcp(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, const double (&Vg)[MV_DD], const double (&Ve)[MV_DD],
                      double (&beta)[MV_D1], double (&Vbeta)[MV_DD]) {
  MV_SHAPE(rt);
  constexpr int CWT = CT > 0 ? (CT > 1 ? CT - 1 : 1) : -MV_CMAX, CWK = mv_mk<CWT>; // the covariate block: order c - 1
  const int CW = C - 1;
  MvBasis<DT> bs;
  MvMoments<DT, CT> m;
  bs.build(Vg, Ve, D);
  mv_pass_moments<DT, CT, Lanes>(g, rt, x, bs, m);
  double sol[DK], vinv[DK], stat = 0.0;
#pragma unroll
  for (int l = 0; l < D; ++l) {
    double Ql[CK * CK];
    mv_unpack_sym<CT>(m.Q + l * T, Ql, C);
    double xPx = Ql[(C - 1) * C + (C - 1)], xPy = m.xHiy[l * C + (C - 1)];
    if (CW > 0) {
      double QW[CWK * CWK], QWi[CWK * CWK];
#pragma unroll
      for (int a = 0; a < CW; ++a)
#pragma unroll
        for (int b = 0; b < CW; ++b) QW[a * CW + b] = Ql[a * C + b];
      mv_spd_inverse<CWT>(QW, QWi, CW);
#pragma unroll
      for (int a = 0; a < CW; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < CW; ++b) s += QWi[a * CW + b] * Ql[b * C + (C - 1)]; // Qi WHix
        xPx -= Ql[a * C + (C - 1)] * s;
        xPy -= s * m.xHiy[l * C + a];
      }
    }
    sol[l] = xPy / xPx;
    vinv[l] = 1.0 / xPx;
    stat += sol[l] * xPy;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < D; ++l) s += bs.UltVeh[l * D + a] * sol[l];
    beta[a] = s;
#pragma unroll
    for (int b = 0; b < D; ++b) {
      double v = 0.0;
#pragma unroll
      for (int l = 0; l < D; ++l) v += bs.UltVeh[l * D + a] * vinv[l] * bs.UltVeh[l * D + b];
      Vbeta[a * D + b] = v;
    }
  }
  return mv_chisq_Q(stat, D);
}

template <int DT, int CT, class Lanes>
MV_OUTLINE double mv_calcp_out(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, const double (&Vg)[MV_DD],
                               const double (&Ve)[MV_DD], double (&beta)[MV_D1], double (&Vbeta)[MV_DD]) {
  return mv_calcp<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
}
template <int DT, int CT, class Lanes>
MV_HD double mv_calcp_sel(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, const double (&Vg)[MV_DD],
                          const double (&Ve)[MV_DD], double (&beta)[MV_D1], double (&Vbeta)[MV_DD]) {
  if constexpr (DT == 0) return mv_calcp_out<0, 0, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
  else return mv_calcp<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
}

// ---------------------------------------------------------------- MphNR (src/mvlmm.cpp:2608-2760)
template <class NR, int N2> MV_OUTLINE double mv_nr_out(NR &nr, bool reml, double lndet_xxt, double (&Vg)[N2], double (&Ve)[N2], int max_iter) {
  return nr.run(reml, lndet_xxt, Vg, Ve, max_iter);
}

// index of the pair (a <= b) in a packed upper triangle of order N (GetIndex :1093-1109)
MV_HD constexpr int mv_tri(int a, int b, int N) { return a <= b ? (2 * N - a + 1) * a / 2 + b - a : (2 * N - b + 1) * b / 2 + a - b; }

// Per-wavefront scratch of the Newton-Raphson stage (LDS on the GPU): the moment tables below are independent of the
// derivative direction, so one sweep of 3 d passes serves the whole gradient and Hessian.  With w_l = 1/(delta D_l + 1),
// u = w o (UltVehi y - Btilde x)  (= the rotated (P y)_k), weights delta^a (a = 0: V_e, 1: V_g) and delta^s (s = a1 + a2):
//   W1[a][l]          sum w_l                      UU[a][p<=q]           sum u_p u_q
//   R[a][j][l][q]     sum x_j w_l u_q              S[a][l1<=l2][j1<=j2]  sum x_j1 x_j2 w_l1 w_l2
//   WW[s][l1<=l2]     sum w_l1 w_l2                Y3[s][q][p<=r]        sum u_p w_q u_r
//   S3[s][l][q][j1<=j2] sum x_j1 x_j2 w_l^2 w_q
struct MvNrLayout {
  int T, TD, VS, H2, W1, UU, R, S, WW, Y3, S3, DT, QI, GRAD, HESS, HINV, LU, LUSZ, DOUBLES;
  // DT: rotated directions; LU: the elimination scratch of invert_hessian, afterwards the 8 d x d tables of crt_factors
  constexpr MvNrLayout(int D, int C)
      : T(C * (C + 1) / 2), TD(D * (D + 1) / 2), VS(TD), H2(2 * VS), W1(0), UU(W1 + 2 * D), R(UU + 2 * TD), S(R + 2 * C * D * D),
        WW(S + 2 * TD * T), Y3(WW + 3 * TD), S3(Y3 + 3 * D * TD), DT(S3 + 3 * D * D * T), QI(DT + VS * D * D), GRAD(QI + D * C * C),
        HESS(GRAD + H2), HINV(HESS + H2 * H2), LU(HINV + H2 * H2), LUSZ((H2 * H2 > 8 * D * D) ? H2 * H2 : 8 * D * D), DOUBLES(LU + LUSZ) {}
};
template <int D, int C> struct MvNrScratch { // the fixed kernels' LDS array
  static constexpr int DOUBLES = MvNrLayout(D, C).DOUBLES;
};

MV_HD void mv_lane_fence() {
#ifdef __HIPCC__
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

template <int DT, int CT, class Lanes> struct MvNr {
  const MvArgs &g;
  double *lds;            // sc.DOUBLES doubles owned by this wavefront (LDS for the fixed kernels, global memory for the run-time one)
  const MvRt rt;
  const MvNrLayout sc;
  const double *x = nullptr;
  MV_HD MvNr(const MvArgs &g_, double *lds_, const MvRt &rt_ = MvRt())
      : g(g_), lds(lds_), rt(rt_), sc(DT > 0 ? DT : rt_.d, CT > 0 ? CT : rt_.c) {}

  // logl at (Vg, Ve) and, if want_dev, gradient + CalcDev's Hessian in lds[GRAD], lds[HESS]
  MV_HD double eval(bool reml, double logl_const, const double (&Vg)[MV_DD], const double (&Ve)[MV_DD], bool want_dev) {
    MV_SHAPE(rt);
    const int n = g.n;
    MvBasis<DT> bs;
    MvMoments<DT, CT> m;
    bs.build(Vg, Ve, D);
    mv_pass_moments<DT, CT, Lanes>(g, rt, x, bs, m);
    double logdet_Q = 0.0, quad = 0.0, Bt[DK * CK], Qi[DK][CK * CK];
#pragma unroll
    for (int l = 0; l < D; ++l) {
      double Ql[CK * CK];
      mv_unpack_sym<CT>(m.Q + l * T, Ql, C);
      logdet_Q += mv_spd_inverse<mv_mt<CT, MV_CMAX>>(Ql, Qi[l], C);
#pragma unroll
      for (int a = 0; a < C; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < C; ++b) s += Qi[l][a * C + b] * m.xHiy[l * C + b];
        Bt[l * C + a] = s;
        quad += s * m.xHiy[l * C + a];
      }
    }
    double logl = logl_const - 0.5 * (m.ll - quad) - 0.5 * (double)n * bs.logdet_Ve;
    if (reml) logl += -0.5 * (logdet_Q - (double)C * bs.logdet_Ve);
    if (!want_dev) return logl;

    // ---- moment passes: one per (power of delta, first component index)
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int l1 = 0; l1 < D; ++l1) {
        double aW1 = 0.0, aUU[TDK], aR[CK * DK], aS[DK * TK], aWW[DK], aY3[TDK], aS3[DK * TK];
#pragma unroll
        for (int i = 0; i < TD; ++i) aUU[i] = aY3[i] = 0.0;
#pragma unroll
        for (int i = 0; i < C * D; ++i) aR[i] = 0.0;
#pragma unroll
        for (int i = 0; i < D * T; ++i) aS[i] = aS3[i] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) aWW[i] = 0.0;
        for (int k = Lanes::lane(); k < n; k += Lanes::N) {
          const double delta = g.eval[k];
          const double ws = (s == 0) ? 1.0 : (s == 1 ? delta : delta * delta);
          double xv[CK], yv[DK], w[DK], u[DK];
          mv_load_x<CT>(g, rt, x, k, xv);
#pragma unroll
          for (int i = 0; i < D; ++i) yv[i] = g.Yt[(long)i * n + k];
#pragma unroll
          for (int l = 0; l < D; ++l) {
            w[l] = 1.0 / (delta * bs.Dl[l] + 1.0);
            double yt = 0.0, bx = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) yt += bs.UltVehi[l * D + i] * yv[i];
#pragma unroll
            for (int j = 0; j < C; ++j) bx += Bt[l * C + j] * xv[j];
            u[l] = w[l] * (yt - bx);
          }
          const double wl = ws * w[l1];
          aW1 += wl;
          if (l1 == 0) {
            int t = 0;
#pragma unroll
            for (int p = 0; p < D; ++p)
#pragma unroll
              for (int q = p; q < D; ++q, ++t) aUU[t] += ws * u[p] * u[q];
          }
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int q = 0; q < D; ++q) aR[j * D + q] += wl * xv[j] * u[q];
          {
            int t = 0;
#pragma unroll
            for (int p = 0; p < D; ++p)
#pragma unroll
              for (int r = p; r < D; ++r, ++t) aY3[t] += wl * u[p] * u[r];
          }
#pragma unroll
          for (int l2 = 0; l2 < D; ++l2) {
            const double ww = wl * w[l2];
            aWW[l2] += ww;          // only l2 >= l1 is stored
            int t = 0;
#pragma unroll
            for (int j1 = 0; j1 < C; ++j1)
#pragma unroll
              for (int j2 = j1; j2 < C; ++j2, ++t) {
                const double xx = xv[j1] * xv[j2];
                aS[l2 * T + t] += ww * xx;              // x x w_l1 w_l2
                aS3[l2 * T + t] += ww * w[l1] * xx;     // x x w_l1^2 w_q   (q = l2)
              }
          }
        }
        // reduce and store (every lane holds the totals; lane 0 writes)
        const bool wr = Lanes::lane() == 0;
        if (s < 2) {
          const double v = Lanes::sum(aW1);
          if (wr) lds[sc.W1 + s * D + l1] = v;
          if (l1 == 0)
#pragma unroll
            for (int i = 0; i < TD; ++i) {
              const double v2 = Lanes::sum(aUU[i]);
              if (wr) lds[sc.UU + s * TD + i] = v2;
            }
#pragma unroll
          for (int j = 0; j < C; ++j)
#pragma unroll
            for (int q = 0; q < D; ++q) {
              const double v2 = Lanes::sum(aR[j * D + q]);
              if (wr) lds[sc.R + ((s * C + j) * D + l1) * D + q] = v2;
            }
#pragma unroll
          for (int l2 = l1; l2 < D; ++l2)
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const double v2 = Lanes::sum(aS[l2 * T + t]);
              if (wr) lds[sc.S + (s * TD + mv_tri(l1, l2, D)) * T + t] = v2;
            }
        }
#pragma unroll
        for (int l2 = l1; l2 < D; ++l2) {
          const double v2 = Lanes::sum(aWW[l2]);
          if (wr) lds[sc.WW + s * TD + mv_tri(l1, l2, D)] = v2;
        }
#pragma unroll
        for (int i = 0; i < TD; ++i) {
          const double v2 = Lanes::sum(aY3[i]);
          if (wr) lds[sc.Y3 + (s * D + l1) * TD + i] = v2;
        }
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double v2 = Lanes::sum(aS3[q * T + t]);
            if (wr) lds[sc.S3 + ((s * D + l1) * D + q) * T + t] = v2;
          }
      }
    // rotated directions and the Q blocks
    if (Lanes::lane() == 0) {
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
          const int v = mv_tri(i, j, D);
#pragma unroll
          for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = 0; q < D; ++q) {
              const double a = bs.UltVehi[p * D + i] * bs.UltVehi[q * D + j];
              lds[sc.DT + (v * D + p) * D + q] = (i == j) ? a : a + bs.UltVehi[p * D + j] * bs.UltVehi[q * D + i];
            }
        }
#pragma unroll
      for (int l = 0; l < D; ++l)
#pragma unroll
        for (int t = 0; t < C * C; ++t) lds[sc.QI + l * C * C + t] = Qi[l][t];
    }
    mv_lane_fence();
    contract(reml);
    mv_lane_fence();
    return logl;
  }

  // symmetric c x c block stored as a packed triangle
  MV_HD double symget(const double *tri, int a, int b) const { return tri[mv_tri(a, b, CT > 0 ? CT : rt.c)]; }

  // gradient and Hessian from the tables (every lane computes the same values; plain loops over LDS)
  MV_HD void contract(bool reml) {
    MV_SHAPE(rt);
    const double *Dtab = lds + sc.DT, *QI = lds + sc.QI;
    double *grad = lds + sc.GRAD, *hess = lds + sc.HESS;
    for (int v = 0; v < VS; ++v) {
      const double *Dv = Dtab + v * D * D;
      for (int a = 0; a < 2; ++a) {
        double yPDPy = 0.0, trHiD = 0.0, trQM = 0.0;
        for (int p = 0; p < D; ++p)
          for (int q = 0; q < D; ++q) yPDPy += Dv[p * D + q] * lds[sc.UU + a * TD + mv_tri(p, q, D)];
        for (int l = 0; l < D; ++l) {
          trHiD += Dv[l * D + l] * lds[sc.W1 + a * D + l];
          if (reml) {
            const double *Sl = lds + sc.S + (a * TD + mv_tri(l, l, D)) * T;
            double t = 0.0;
            for (int j1 = 0; j1 < C; ++j1)
              for (int j2 = 0; j2 < C; ++j2) t += QI[l * C * C + j1 * C + j2] * symget(Sl, j1, j2);
            trQM += Dv[l * D + l] * t;
          }
        }
        grad[(a ? 0 : VS) + v] = -0.5 * (trHiD - trQM) + 0.5 * yPDPy;
      }
    }
    for (int i = 0; i < H2 * H2; ++i) hess[i] = 0.0;
    for (int v1 = 0; v1 < VS; ++v1)
      for (int v2 = v1; v2 < VS; ++v2) {
        const double *D1 = Dtab + v1 * D * D, *D2 = Dtab + v2 * D * D;
        double dev2[3];
        for (int sI = 0; sI < 3; ++sI) { // ee, ge (D1 = V_g direction, D2 = V_e direction), gg
          const int a1 = sI >= 1, a2 = sI == 2;
          double yy = 0.0, trHH = 0.0, t2 = 0.0, t4 = 0.0, rQr = 0.0;
          for (int q = 0; q < D; ++q) {
            const double *Y3q = lds + sc.Y3 + (sI * D + q) * TD;
            for (int p = 0; p < D; ++p)
              for (int r = 0; r < D; ++r) yy += D1[q * D + p] * D2[q * D + r] * Y3q[mv_tri(p, r, D)];
          }
          for (int l = 0; l < D; ++l) {
            double r1[CK], r2[CK];
            for (int j = 0; j < C; ++j) {
              double s1 = 0.0, s2 = 0.0;
              for (int q = 0; q < D; ++q) {
                s1 += D1[l * D + q] * lds[sc.R + ((a1 * C + j) * D + l) * D + q];
                s2 += D2[l * D + q] * lds[sc.R + ((a2 * C + j) * D + l) * D + q];
              }
              r1[j] = s1;
              r2[j] = s2;
            }
            for (int j1 = 0; j1 < C; ++j1)
              for (int j2 = 0; j2 < C; ++j2) rQr += r1[j1] * QI[l * C * C + j1 * C + j2] * r2[j2];
          }
          for (int l1 = 0; l1 < D; ++l1)
            for (int l2 = 0; l2 < D; ++l2) {
              const double dd = D1[l1 * D + l2] * D2[l2 * D + l1];
              trHH += dd * lds[sc.WW + sI * TD + mv_tri(l1, l2, D)];
              if (reml) {
                const double *S3 = lds + sc.S3 + ((sI * D + l1) * D + l2) * T; // q = l2
                double t = 0.0;
                for (int j1 = 0; j1 < C; ++j1)
                  for (int j2 = 0; j2 < C; ++j2) t += QI[l1 * C * C + j1 * C + j2] * symget(S3, j1, j2);
                t2 += dd * t;
                const double *A = lds + sc.S + (a1 * TD + mv_tri(l1, l2, D)) * T;
                const double *Bm = lds + sc.S + (a2 * TD + mv_tri(l1, l2, D)) * T;
                double tr = 0.0; // tr(Qi_l1 A Qi_l2 B)
                for (int i1 = 0; i1 < C; ++i1)
                  for (int i2 = 0; i2 < C; ++i2) {
                    double qa = 0.0, qb = 0.0; // (Qi_l1 A)[i1][i2], (Qi_l2 B)[i2][i1]
                    for (int t3 = 0; t3 < C; ++t3) {
                      qa += QI[l1 * C * C + i1 * C + t3] * symget(A, t3, i2);
                      qb += QI[l2 * C * C + i2 * C + t3] * symget(Bm, t3, i1);
                    }
                    tr += qa * qb;
                  }
                t4 += dd * tr;
              }
            }
          double tr = trHH;
          if (reml) tr += -2.0 * t2 + t4;
          dev2[sI] = 0.5 * tr - (yy - rQr);
        }
        hess[v1 * H2 + v2] = hess[v2 * H2 + v1] = dev2[2];
        hess[(v1 + VS) * H2 + v2 + VS] = hess[(v2 + VS) * H2 + v1 + VS] = dev2[0];
        hess[v1 * H2 + v2 + VS] = hess[(v2 + VS) * H2 + v1] = dev2[1]; // mirrored as :2494-2504 does
        hess[v2 * H2 + v1 + VS] = hess[(v1 + VS) * H2 + v2] = dev2[1];
      }
  }

  // Hinv = HESS^-1 by LU with partial pivoting (LUDecomp / LUInvert :2510-2518), in scratch
  MV_HD void invert_hessian() {
    MV_SHAPE(rt);
    double *lu = lds + sc.LU, *Hi = lds + sc.HINV;
    const double *H = lds + sc.HESS;
    for (int i = 0; i < H2 * H2; ++i) lu[i] = H[i];
    int perm[H2K];
    for (int i = 0; i < H2; ++i) perm[i] = i;
    for (int j = 0; j < H2; ++j) {
      int ip = j;
      double amax = fabs(lu[j * H2 + j]);
      for (int i = j + 1; i < H2; ++i)
        if (fabs(lu[i * H2 + j]) > amax) {
          amax = fabs(lu[i * H2 + j]);
          ip = i;
        }
      if (ip != j) {
        for (int k = 0; k < H2; ++k) {
          const double t = lu[j * H2 + k];
          lu[j * H2 + k] = lu[ip * H2 + k];
          lu[ip * H2 + k] = t;
        }
        const int t = perm[j];
        perm[j] = perm[ip];
        perm[ip] = t;
      }
      const double ajj = lu[j * H2 + j];
      if (ajj != 0.0)
        for (int i = j + 1; i < H2; ++i) {
          const double f = lu[i * H2 + j] / ajj;
          lu[i * H2 + j] = f;
          for (int k = j + 1; k < H2; ++k) lu[i * H2 + k] -= f * lu[j * H2 + k];
        }
    }
    for (int col = 0; col < H2; ++col) {
      double xs[H2K];
      for (int i = 0; i < H2; ++i) xs[i] = (perm[i] == col) ? 1.0 : 0.0;
      for (int i = 0; i < H2; ++i)
        for (int k = 0; k < i; ++k) xs[i] -= lu[i * H2 + k] * xs[k];
      for (int ii = H2 - 1; ii >= 0; --ii) {
        for (int k = ii + 1; k < H2; ++k) xs[ii] -= lu[ii * H2 + k] * xs[k];
        xs[ii] /= lu[ii * H2 + ii];
      }
      for (int i = 0; i < H2; ++i) Hi[i * H2 + col] = xs[i];
    }
  }

  // CalcCRT (src/mvlmm.cpp:2054-2331) at the point of the last eval(..., true) + invert_hessian(): Rothenberg's Edgeworth
  // correction factors.  The reference forms dense dc x dc products of Qi, xHiDHix (M) and xHiDHiDHix (MM) and takes traces of
  // their SNP blocks against the inverse of Qi's SNP block; every one of those traces is invariant under the rotation that
  // makes H_k diagonal, and there Qi is block-separable per component l (QI[l], c x c), its SNP block is diag(q_l),
  // q_l = QI[l][z][z] (z = c - 1), M_v[(i,l1),(j,l2)] = Dt_v[l1][l2] S[a][l1,l2][i,j] and the diagonal blocks of MM are
  // sum_q Dt_1[l][q] Dt_2[q][l] S3[s][l][q][i,j] -- the tables the Hessian already uses.  With
  //   r^a[l1][l2][j] = sum_i QI[l1][z][i] S[a][l1,l2][i][j]
  //   g^a[l1][l2]    = sum_j r^a[l1][l2][j] QI[l2][j][z]                         (SNP block of Qi M Qi = Dt o g^a)
  //   h^{ab}[l][q]   = sum_ij r^a[l][q][i] QI[q][i][j] r^b[l][q][j]              (diagonal of the SNP block of Qi M Qi M Qi)
  //   k^s[l][q]      = sum_ij QI[l][z][i] S3[s][l][q][i][j] QI[l][j][z]          (diagonal of the SNP block of Qi MM Qi)
  // the reference's trC, trCC, trB follow as sums over (l, q); B, C, D and crt_a, b, c as written there (:2303-2331).
  double crt[3] = {0.0, 0.0, 0.0};
  MV_HD void crt_factors() {
    MV_SHAPE(rt);
    const int z = C - 1, DD = D * D;
    const double *QI = lds + sc.QI, *Dtab = lds + sc.DT, *Hi = lds + sc.HINV;
    double *gt = lds + sc.LU, *ht = gt + 2 * DD, *kt = ht + 3 * DD; // g[a], h[gg, ge, ee], k[ee, ge, gg]
    for (int l1 = 0; l1 < D; ++l1)
      for (int l2 = 0; l2 < D; ++l2) {
        double r[2][CK];
        for (int a = 0; a < 2; ++a) {
          const double *Sa = lds + sc.S + (a * TD + mv_tri(l1, l2, D)) * T;
          for (int j = 0; j < C; ++j) {
            double t = 0.0;
            for (int i = 0; i < C; ++i) t += QI[l1 * C * C + z * C + i] * symget(Sa, i, j);
            r[a][j] = t;
          }
          double gg = 0.0;
          for (int j = 0; j < C; ++j) gg += r[a][j] * QI[l2 * C * C + j * C + z];
          gt[a * DD + l1 * D + l2] = gg;
        }
        double hgg = 0.0, hge = 0.0, hee = 0.0;
        for (int i = 0; i < C; ++i)
          for (int j = 0; j < C; ++j) {
            const double qq = QI[l2 * C * C + i * C + j];
            hgg += r[1][i] * qq * r[1][j];
            hge += r[1][i] * qq * r[0][j];
            hee += r[0][i] * qq * r[0][j];
          }
        ht[0 * DD + l1 * D + l2] = hgg;
        ht[1 * DD + l1 * D + l2] = 2.0 * hge; // Qi M_g Qi M_e Qi + Qi M_e Qi M_g Qi: equal on this diagonal (QI symmetric)
        ht[2 * DD + l1 * D + l2] = hee;
        for (int sI = 0; sI < 3; ++sI) {
          const double *S3 = lds + sc.S3 + ((sI * D + l1) * D + l2) * T;
          double t = 0.0;
          for (int i = 0; i < C; ++i)
            for (int j = 0; j < C; ++j) t += QI[l1 * C * C + z * C + i] * symget(S3, i, j) * QI[l1 * C * C + j * C + z];
          kt[sI * DD + l1 * D + l2] = t;
        }
      }
    mv_lane_fence();
    double qinv[DK];
    for (int l = 0; l < D; ++l) qinv[l] = 1.0 / QI[l * C * C + z * C + z];
    double Bs = 0.0, Cs = 0.0, Ds = 0.0;
    for (int v1 = 0; v1 < VS; ++v1) {
      const double *D1 = Dtab + v1 * DD;
      double trCg1 = 0.0, trCe1 = 0.0;
      for (int l = 0; l < D; ++l) {
        trCg1 -= D1[l * D + l] * gt[DD + l * D + l] * qinv[l];
        trCe1 -= D1[l * D + l] * gt[l * D + l] * qinv[l];
      }
      for (int v2 = v1; v2 < VS; ++v2) {
        const double *D2 = Dtab + v2 * DD;
        double trCg2 = 0.0, trCe2 = 0.0, trCC_gg = 0.0, trCC_ge = 0.0, trCC_ee = 0.0, trB_gg = 0.0, trB_ge = 0.0, trB_ee = 0.0;
        for (int l = 0; l < D; ++l) {
          trCg2 -= D2[l * D + l] * gt[DD + l * D + l] * qinv[l];
          trCe2 -= D2[l * D + l] * gt[l * D + l] * qinv[l];
        }
        for (int l1 = 0; l1 < D; ++l1)
          for (int l2 = 0; l2 < D; ++l2) {
            const double dd = D1[l1 * D + l2] * D2[l2 * D + l1];
            const double w12 = dd * qinv[l1] * qinv[l2];
            const double g1 = gt[DD + l1 * D + l2], g0 = gt[l1 * D + l2], g1t = gt[DD + l2 * D + l1], g0t = gt[l2 * D + l1];
            trCC_gg += w12 * g1 * g1t;
            trCC_ge += w12 * (g1 * g0t + g0 * g1t);
            trCC_ee += w12 * g0 * g0t;
            const double wl = dd * qinv[l1]; // (l, q) = (l1, l2)
            trB_gg += wl * (kt[2 * DD + l1 * D + l2] - ht[0 * DD + l1 * D + l2]);
            trB_ge += wl * (2.0 * kt[1 * DD + l1 * D + l2] - ht[1 * DD + l1 * D + l2]);
            trB_ee += wl * (kt[0 * DD + l1 * D + l2] - ht[2 * DD + l1 * D + l2]);
          }
        const double trD_gg = 2.0 * trB_gg, trD_ge = 2.0 * trB_ge, trD_ee = 2.0 * trB_ee;
        const double h_gg = -Hi[v1 * H2 + v2], h_ge = -Hi[v1 * H2 + v2 + VS], h_ee = -Hi[(v1 + VS) * H2 + v2 + VS];
        const double f = (v1 != v2) ? 2.0 : 1.0;
        Bs += f * (h_gg * trB_gg + h_ge * trB_ge + h_ee * trB_ee);
        Cs += f * (h_gg * (trCC_gg + 0.5 * trCg1 * trCg2) + h_ge * (trCC_ge + 0.5 * trCg1 * trCe2 + 0.5 * trCe1 * trCg2) +
                   h_ee * (trCC_ee + 0.5 * trCe1 * trCe2));
        Ds += f * (h_gg * (trCC_gg + 0.5 * trD_gg) + h_ge * (trCC_ge + 0.5 * trD_ge) + h_ee * (trCC_ee + 0.5 * trD_ee));
      }
    }
    crt[0] = 2.0 * Ds - Cs;
    crt[1] = 2.0 * Bs;
    crt[2] = Cs;
  }

  MV_HD static bool is_pd(const double (&V)[MV_DD], int d) {
    constexpr int DK = mv_dk<DT>;
    const int D = DT > 0 ? DT : d;
    double w[DK], Z[DK * DK];
    mv_jacobi<mv_mt<DT, MV_DMAX>>(V, w, Z, D);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) ok = ok && (w[i] > 0);
    return ok;
  }

  // MphNR with the per-SNP limits (nr_iter / 10, nr_prec * 10); returns logl_H1
  // max_iter < 0: g.nr_iter.  With g.crt the factors of the LAST CalcDev call stay in crt[] (:2522-2530, for 'R' and 'L' alike)
  MV_HD double operator()(bool reml, double lndet_xxt, double (&Vg)[MV_DD], double (&Ve)[MV_DD], int max_iter = -1) {
    if constexpr (DT == 0) return mv_nr_out(*this, reml, lndet_xxt, Vg, Ve, max_iter);
    else return run(reml, lndet_xxt, Vg, Ve, max_iter);
  }
  MV_HD double run(bool reml, double lndet_xxt, double (&Vg)[MV_DD], double (&Ve)[MV_DD], int max_iter) {
    MV_SHAPE(rt);
    constexpr double LOG2PI = 1.8378770664093454836;
    const int n = g.n;
    const double logl_const = reml ? -0.5 * (double)(n - C) * (double)D * LOG2PI + 0.5 * (double)D * lndet_xxt
                                   : -0.5 * (double)n * (double)D * LOG2PI;
    double Vg_save[DK * DK], Ve_save[DK * DK];
    double logl_old = 0.0, logl_new = 0.0;
    const double *Hi = lds + sc.HINV, *grad = lds + sc.GRAD;
    const int iters = max_iter < 0 ? g.nr_iter : max_iter;
    crt[0] = crt[1] = crt[2] = 0.0;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
      for (int i = 0; i < D * D; ++i) {
        Vg_save[i] = Vg[i];
        Ve_save[i] = Ve[i];
      }
      double step_scale = 1.0;
      int step_iter = 0;
      bool flag_pd;
      do {
#pragma unroll
        for (int i = 0; i < D * D; ++i) {
          Vg[i] = Vg_save[i];
          Ve[i] = Ve_save[i];
        }
        if (t != 0) { // UpdateVgVe :2557-2606
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) {
              const int v = mv_tri(i, j, D);
              double sg = 0.0, se = 0.0;
              for (int q = 0; q < H2; ++q) {
                sg += Hi[v * H2 + q] * grad[q];
                se += Hi[(v + VS) * H2 + q] * grad[q];
              }
              Vg[i * D + j] = Vg[j * D + i] = Vg_save[i * D + j] - step_scale * sg;
              Ve[i * D + j] = Ve[j * D + i] = Ve_save[i * D + j] - step_scale * se;
            }
        }
        flag_pd = is_pd(Ve, D) && is_pd(Vg, D);
        if (flag_pd) logl_new = eval(reml, logl_const, Vg, Ve, false);
        step_scale /= 2.0;
        step_iter++;
      } while ((!flag_pd || logl_new < logl_old || logl_new - logl_old > 10) && step_iter < 10 && t != 0);
      if (t != 0) {
        if (logl_new < logl_old || !flag_pd) {
#pragma unroll
          for (int i = 0; i < D * D; ++i) {
            Vg[i] = Vg_save[i];
            Ve[i] = Ve_save[i];
          }
          break;
        }
        if (logl_new - logl_old < g.nr_prec) break;
      }
      logl_old = logl_new;
      eval(reml, logl_const, Vg, Ve, true);
      invert_hessian();
      mv_lane_fence();
      if (g.crt && C > 1) {
        crt_factors();
        mv_lane_fence();
      }
    }
    return logl_new;
  }
};

// One SNP: the body of the loop at src/mvlmm.cpp:3287-3374 (with -crt: PCRT on the SNPs that reach MphNR).  nr: callable (reml) -> logl_H1 that refines
// Vg, Ve by Newton-Raphson, or a no-op returning NaN when the stage is not compiled in.
// B = GLS estimate of the fixed effects at (Vg, Ve): what MphCalcBeta (:835-935) leaves in B
template <int DT, int CT, class Lanes>
MV_HD void mv_gls_B(const MvArgs &g, const MvRt &rt, const double *__restrict__ x, const double (&Vg)[MV_DD], const double (&Ve)[MV_DD],
                    double (&B)[MV_DC]) {
  MV_SHAPE(rt);
  MvBasis<DT> bs;
  MvMoments<DT, CT> m;
  bs.build(Vg, Ve, D);
  mv_pass_moments<DT, CT, Lanes>(g, rt, x, bs, m);
  double bl[DK * CK];
#pragma unroll
  for (int l = 0; l < D; ++l) {
    double Ql[CK * CK], Qi[CK * CK];
    mv_unpack_sym<CT>(m.Q + l * T, Ql, C);
    mv_spd_inverse<mv_mt<CT, MV_CMAX>>(Ql, Qi, C);
#pragma unroll
    for (int a = 0; a < C; ++a) {
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < C; ++b) s += Qi[a * C + b] * m.xHiy[l * C + b];
      bl[l * C + a] = s;
    }
  }
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < D; ++l) s += bs.UltVeh[l * D + a] * bl[l * C + j];
      B[a * C + j] = s;
    }
}

// the record of one SNP: beta[d], Vbeta[v], Vg[v], Ve[v], p_wald, p_lrt, p_score (MPHSUMSTAT, src/param.h:68-77)
template <int DT>
MV_HD void mv_store_snp(const MvArgs &g, long s, int D, const double (&beta)[MV_D1], const double (&Vbeta)[MV_DD],
                        const double (&Vg)[MV_DD], const double (&Ve)[MV_DD], double p_wald, double p_lrt, double p_score) {
  const int V = D * (D + 1) / 2;
  double *o = g.out + s * g.stride;
#pragma unroll
  for (int i = 0; i < D; ++i) o[i] = beta[i];
  int q = 0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j, ++q) {
      o[D + q] = Vbeta[i * D + j];
      o[D + V + q] = Vg[i * D + j];
      o[D + 2 * V + q] = Ve[i * D + j];
    }
  o[D + 3 * V] = p_wald;
  o[D + 3 * V + 1] = p_lrt;
  o[D + 3 * V + 2] = p_score;
}

template <int DT, int CT, class Lanes, class NR>
MV_HD void mv_one_snp(const MvArgs &g, long s, NR &&nr) {
  const MvRt &rt = nr.rt;
  MV_SHAPE(rt);
  const int CW = C - 1;
  const double *__restrict__ x = g.UtX + s * g.ld;
  double Vg[DK * DK], Ve[DK * DK], Vg0[DK * DK], Ve0[DK * DK], B[DK * CK], beta[DK], Vbeta[DK * DK], XXti[CK * CK];
#pragma unroll
  for (int i = 0; i < D * D; ++i) {
    Vg[i] = Vg0[i] = g.Vg_null[i];
    Ve[i] = Ve0[i] = g.Ve_null[i];
    Vbeta[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    beta[i] = 0.0;
#pragma unroll
    for (int j = 0; j < CW; ++j) B[i * C + j] = g.B_null[i * CW + j];
    B[i * C + CW] = 0.0;
  }
  const double lndet_xxt = mv_xxt<CT, Lanes>(g, rt, x, XXti);
  double p_wald = 0.0, p_lrt = 0.0, p_score = 0.0;
  if (g.a_mode == 3 || g.a_mode == 4) {
    p_score = mv_calcp_sel<DT, CT, Lanes>(g, rt, x, Vg0, Ve0, beta, Vbeta);
    if (p_score < g.p_nr && g.crt == 1) { // :3302-3306: one CalcDev at the null estimates
      nr(true, lndet_xxt, Vg, Ve, 1);
      p_score = mv_pcrt(3, D, p_score, nr.crt);
    }
  }
  if (g.a_mode == 2 || g.a_mode == 4) {
    double logl_H1 = mv_em_sel<DT, CT, Lanes>(g, rt, x, false, g.em_iter, g.em_prec, lndet_xxt, XXti, Vg, Ve, B);
    mv_calcp_sel<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
    p_lrt = mv_chisq_Q(2.0 * (logl_H1 - g.logl_H0), D);
    if (p_lrt < g.p_nr) {
      logl_H1 = nr(false, lndet_xxt, Vg, Ve);
      mv_calcp_sel<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
      p_lrt = mv_chisq_Q(2.0 * (logl_H1 - g.logl_H0), D);
      if (g.crt == 1) p_lrt = mv_pcrt(2, D, p_lrt, nr.crt);
    }
  }
  if (g.a_mode == 1 || g.a_mode == 4) {
    mv_em_sel<DT, CT, Lanes>(g, rt, x, true, g.em_iter, g.em_prec, lndet_xxt, XXti, Vg, Ve, B);
    p_wald = mv_calcp_sel<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
    if (p_wald < g.p_nr) {
      nr(true, lndet_xxt, Vg, Ve);
      p_wald = mv_calcp_sel<DT, CT, Lanes>(g, rt, x, Vg, Ve, beta, Vbeta);
      if (g.crt == 1) p_wald = mv_pcrt(1, D, p_wald, nr.crt);
    }
  }
  if (Lanes::lane() == 0) mv_store_snp<DT>(g, s, D, beta, Vbeta, Vg, Ve, p_wald, p_lrt, p_score);
}

// One SNP of the gene-environment interaction test, the loop body at src/mvlmm.cpp:4259-4353 (AnalyzeBimbamGXE; AnalyzePlinkGXE
// :4416 is the same after the read): X = (W, env, x, x o env) with c rows, tested row = the interaction.  The NULL of the test
// holds the SNP's main effect, so it is fitted per SNP on the first c - 1 rows (REML for the score / Wald statistics, ML for the
// LRT reference, each EM + Newton-Raphson with the per-SNP limits and carrying V_g, V_e on from one fit to the next as the
// reference's loop does); g.B_null is d x (c - 2), the columns of the global null fit.  Run-time instance only.
template <class Lanes>
MV_HD void mv_one_snp_gxe(const MvArgs &g, long s, double *scratch) {
  constexpr int DT = 0, CT = 0;
  MvRt rt1, rt0;                  // alternative (c rows, two SNP rows), per-SNP null (c - 1 rows, the SNP last)
  rt1.d = rt0.d = g.d;
  rt1.c = g.c;
  rt0.c = g.c - 1;
  const double *__restrict__ x = g.UtX + s * g.ld;
  rt1.x2 = g.UtX2 + s * g.ld;
  MV_SHAPE(rt1);
  const int C0 = C - 1, CW = C - 2;
  MvNr<0, 0, Lanes> nr1(g, scratch, rt1), nr0(g, scratch, rt0);
  nr1.x = nr0.x = x;
  double Vg[DK * DK], Ve[DK * DK], Vg0[DK * DK], Ve0[DK * DK], B[DK * CK], B0[DK * CK], beta[DK], Vbeta[DK * DK], XXti[CK * CK], XXti0[CK * CK];
  for (int i = 0; i < D * D; ++i) {
    Vg[i] = Vg0[i] = g.Vg_null[i];
    Ve[i] = Ve0[i] = g.Ve_null[i];
    Vbeta[i] = 0.0;
  }
  for (int i = 0; i < D; ++i) {
    beta[i] = 0.0;
    for (int j = 0; j < CW; ++j) B[i * C + j] = g.B_null[i * CW + j];
    B[i * C + CW] = B[i * C + CW + 1] = 0.0;
  }
  // B_sub2 is a view of B (d x (c + 1) storage in the reference): the per-SNP null fit and the alternative share its columns
  auto to_sub = [&]() {
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < C0; ++j) B0[i * C0 + j] = B[i * C + j];
  };
  auto from_sub = [&]() {
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < C0; ++j) B[i * C + j] = B0[i * C0 + j];
  };
  const double lndet_xxt = mv_xxt<0, Lanes>(g, rt1, x, XXti), lndet_xxt0 = mv_xxt<0, Lanes>(g, rt0, x, XXti0);
  double p_wald = 0.0, p_lrt = 0.0, p_score = 0.0, logl_H0 = 0.0;
  if (g.a_mode == 3 || g.a_mode == 4) { // :4262-4272
    to_sub();
    mv_em_sel<0, 0, Lanes>(g, rt0, x, true, g.em_iter, g.em_prec, lndet_xxt0, XXti0, Vg, Ve, B0);
    nr0(true, lndet_xxt0, Vg, Ve);
    mv_gls_B<0, 0, Lanes>(g, rt0, x, Vg, Ve, B0); // MphCalcBeta leaves the GLS estimate in B_sub2
    from_sub();
  }
  if (g.a_mode == 2 || g.a_mode == 4) { // :4274-4284
    to_sub();
    mv_em_sel<0, 0, Lanes>(g, rt0, x, false, g.em_iter, g.em_prec, lndet_xxt0, XXti0, Vg, Ve, B0);
    logl_H0 = nr0(false, lndet_xxt0, Vg, Ve);
    mv_gls_B<0, 0, Lanes>(g, rt0, x, Vg, Ve, B0);
    from_sub();
  }
  if (g.a_mode == 3 || g.a_mode == 4) {
    p_score = mv_calcp_sel<0, 0, Lanes>(g, rt1, x, Vg0, Ve0, beta, Vbeta);
    if (p_score < g.p_nr && g.crt == 1) {
      nr1(true, lndet_xxt, Vg, Ve, 1);
      p_score = mv_pcrt(3, D, p_score, nr1.crt);
    }
  }
  if (g.a_mode == 2 || g.a_mode == 4) {
    double logl_H1 = mv_em_sel<0, 0, Lanes>(g, rt1, x, false, g.em_iter, g.em_prec, lndet_xxt, XXti, Vg, Ve, B);
    mv_calcp_sel<0, 0, Lanes>(g, rt1, x, Vg, Ve, beta, Vbeta);
    p_lrt = mv_chisq_Q(2.0 * (logl_H1 - logl_H0), D);
    if (p_lrt < g.p_nr) {
      logl_H1 = nr1(false, lndet_xxt, Vg, Ve);
      mv_calcp_sel<0, 0, Lanes>(g, rt1, x, Vg, Ve, beta, Vbeta);
      p_lrt = mv_chisq_Q(2.0 * (logl_H1 - logl_H0), D);
      if (g.crt == 1) p_lrt = mv_pcrt(2, D, p_lrt, nr1.crt);
    }
  }
  if (g.a_mode == 1 || g.a_mode == 4) {
    mv_em_sel<0, 0, Lanes>(g, rt1, x, true, g.em_iter, g.em_prec, lndet_xxt, XXti, Vg, Ve, B);
    p_wald = mv_calcp_sel<0, 0, Lanes>(g, rt1, x, Vg, Ve, beta, Vbeta);
    if (p_wald < g.p_nr) {
      nr1(true, lndet_xxt, Vg, Ve);
      p_wald = mv_calcp_sel<0, 0, Lanes>(g, rt1, x, Vg, Ve, beta, Vbeta);
      if (g.crt == 1) p_wald = mv_pcrt(1, D, p_wald, nr1.crt);
    }
  }
  if (g.flip && g.flip[s])
    for (int i = 0; i < D; ++i) beta[i] = -beta[i];
  if (Lanes::lane() == 0) mv_store_snp<0>(g, s, D, beta, Vbeta, Vg, Ve, p_wald, p_lrt, p_score);
}

} // namespace gemma_hip

// ---------------------------------------------------------------- null model (src/mvlmm.cpp:3056-3208)
namespace gemma_hip {

struct MvNullArgs {
  MvArgs g;            // eval, Wt (all c covariate rows: the last one plays the "x" row), Yt, n; nr_iter / nr_prec
  int em_iter;
  double em_prec;
  double Vg0[MV_DMAX * MV_DMAX], Ve0[MV_DMAX * MV_DMAX]; // MphInitial's starting point
  double *out;         // 2 x (d*d + d*d + d*c + 1): REMLE then MLE block: Vg, Ve, B (d x c), logl
  // run-time instance: g.d, g.c = the phenotypes and the covariates of THIS fit (c counts the last covariate, which plays the x row)
};

// EM + NR for REML, then for ML starting from the REML fit; one "lane group" does the whole fit
template <int DT, int CT, class Lanes> MV_HD void mv_null_fit(const MvNullArgs &a, double *scratch) {
  const MvArgs &g = a.g;
  MvRt rt;
  rt.d = g.d;
  rt.c = g.c;
  MV_SHAPE(rt);
  const double *x = g.Wt + (long)(C - 1) * g.n;
  double Vg[DK * DK], Ve[DK * DK], B[DK * CK], XXti[CK * CK];
#pragma unroll
  for (int i = 0; i < D * D; ++i) {
    Vg[i] = a.Vg0[i];
    Ve[i] = a.Ve0[i];
  }
#pragma unroll
  for (int i = 0; i < D * C; ++i) B[i] = 0.0;
  const double lndet_xxt = mv_xxt<CT, Lanes>(g, rt, x, XXti);
  MvNr<DT, CT, Lanes> nr(g, scratch, rt);
  nr.x = x;
  const int BLK = 2 * D * D + D * C + 1;
  for (int pass = 0; pass < 2; ++pass) {
    const bool reml = pass == 0;
    mv_em_sel<DT, CT, Lanes>(g, rt, x, reml, a.em_iter, a.em_prec, lndet_xxt, XXti, Vg, Ve, B);
    const double logl = nr(reml, lndet_xxt, Vg, Ve);
    mv_gls_B<DT, CT, Lanes>(g, rt, x, Vg, Ve, B);
    if (Lanes::lane() == 0) {
      double *o = a.out + pass * BLK;
#pragma unroll
      for (int i = 0; i < D * D; ++i) {
        o[i] = Vg[i];
        o[D * D + i] = Ve[i];
      }
#pragma unroll
      for (int i = 0; i < D * C; ++i) o[2 * D * D + i] = B[i];
      o[2 * D * D + D * C] = logl;
    }
  }
}

} // namespace gemma_hip
