/* CPU ORACLE for the multivariate LMM (GEMMA src/mvlmm.cpp) -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's per-SNP multivariate path: EigenProc, CalcQi, the EM of MphEM, the Wald
 * statistic of MphCalcP, the Newton-Raphson of MphNR (gradient / observed Hessian of CalcDev) and MphInitial.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product (gemma_amd/csrc) never does.
 *
 * PINNING: the reference's own tests hold no numeric golden for this path (test/dev_test_suite.sh:196-208 checks a line
 * count; its value checksum is commented out).  It is pinned on the reference ITSELF instead: oracle/_ref (the reference's
 * sources compiled unchanged against oracle/gslshim, oracle/Makefile `ref`) run on issue243 (2 traits) and on issue188's
 * genotypes with 3 simulated traits; this file reproduces every printed digit of beta, Vbeta, p_wald / p_lrt / p_score
 * and the null-model matrices in all four modes for d = 2 and in the REML / score modes for d = 3, and the reference's
 * free functions MphEM / MphNR / MphCalcP called directly (oracle/ref_bridge.cpp) to 1e-9 (tests/test_reference_pin.py).
 * The one exception is the reference's own instability: its ML EM for d >= 3 depends on the eigenvector signs LAPACK
 * happens to return (mv_jacobi below), which flip under 1e-15 perturbations, so on ~10 % of SNPs its -lmm 2/4 trajectory
 * is not reproducible by anything but the same binary on the same machine; there the test bounds the likelihood instead.
 * tests/test_oracle_mvlmm.py keeps the independent legs: the univariate oracle at d = 1 and finite differences of the
 * log-likelihood for the gradient and Hessian.
 * -crt (round 3): CalcCRT is restated in the reference's own dense form (mv_calc_crt) and PCRT with GSL's chi-square quantile
 * iteration; pinned on the reference's crt_a, crt_b, crt_c handed back by its MphNR for 'R' and 'L' (ref_MphNR_crt, 1e-6), on its
 * PCRT in the three modes (1e-8) and on its -crt output for both fixtures (tests/golden/ref_mv_crt.npz).
 *
 * Where the reference expands  P = H^-1 - H^-1 X Q^-1 X^T H^-1  into eight products of precomputed tables
 * (src/mvlmm.cpp:1863-2049), this file evaluates the same quantities from u_k = (P y)_k directly; the algebra is
 * identical (see the derivation next to dev_eval), the rounding is not -- no statement here depends on it.
 *
 * Layout: Y is d x n (trait-major rows), X is c x n (covariate rows; the SNP is the last row), eval has n entries.
 * Small matrices are row-major with their natural leading dimension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MV_MAXD 8
#define MV_MAXC 12
#define MV_MAXDC (MV_MAXD * MV_MAXC)
#define MV_MAXV (MV_MAXD * (MV_MAXD + 1) / 2)

/* from gemma_oracle.c (same shared object) */
void orc_CalcLambda_null(char func_name, size_t n, size_t c, const double *eval, const double *UtW, const double *Uty,
                         double l_min, double l_max, size_t n_region, double *lambda, double *logl_H0);
void orc_CalcLmmVgVeBeta(size_t n, size_t c, const double *eval, const double *UtW, const double *Uty, double lambda,
                         double *vg, double *ve, double *beta, double *se_beta);

/* ------------------------------------------------------------------ small dense helpers */
/* cyclic Jacobi for a symmetric m x m matrix: A = V diag(w) V^T, w ascending, V[:, i] the i-th vector, signed so
 * that its entry of largest magnitude is positive.  (the reference calls LAPACK through EigenDecomp(.., 0),
 * src/lapack.cpp:173-266.  Every use is invariant to the signs and the order EXCEPT the ML EM, which subtracts the
 * previous iteration's UltVehiBX from this iteration's UltVehiY (:679-686): there only jumps between two consecutive,
 * nearly equal matrices matter, and a fixed convention avoids them; LAPACK's own signs are not reproducible.) */
static void mv_jacobi(const double *A, size_t m, double *w, double *V) {
  double a[MV_MAXD * MV_MAXD];
  memcpy(a, A, m * m * sizeof(double));
  for (size_t i = 0; i < m; ++i)
    for (size_t j = 0; j < m; ++j) V[i * m + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (size_t i = 0; i < m; ++i) {
      diag += a[i * m + i] * a[i * m + i];
      for (size_t j = i + 1; j < m; ++j) off += a[i * m + j] * a[i * m + j];
    }
    if (off <= 1e-34 * diag || off == 0.0) break;
    for (size_t p = 0; p < m; ++p)
      for (size_t q = p + 1; q < m; ++q) {
        const double apq = a[p * m + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * m + q] - a[p * m + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (size_t k = 0; k < m; ++k) { /* columns p, q */
          const double akp = a[k * m + p], akq = a[k * m + q];
          a[k * m + p] = cs * akp - sn * akq;
          a[k * m + q] = sn * akp + cs * akq;
        }
        for (size_t k = 0; k < m; ++k) { /* rows p, q */
          const double apk = a[p * m + k], aqk = a[q * m + k];
          a[p * m + k] = cs * apk - sn * aqk;
          a[q * m + k] = sn * apk + cs * aqk;
        }
        for (size_t k = 0; k < m; ++k) {
          const double vkp = V[k * m + p], vkq = V[k * m + q];
          V[k * m + p] = cs * vkp - sn * vkq;
          V[k * m + q] = sn * vkp + cs * vkq;
        }
      }
  }
  for (size_t i = 0; i < m; ++i) w[i] = a[i * m + i];
  for (size_t i = 0; i + 1 < m; ++i) { /* ascending, columns follow */
    size_t mi = i;
    for (size_t j = i + 1; j < m; ++j)
      if (w[j] < w[mi]) mi = j;
    if (mi != i) {
      double t = w[i];
      w[i] = w[mi];
      w[mi] = t;
      for (size_t k = 0; k < m; ++k) {
        t = V[k * m + i];
        V[k * m + i] = V[k * m + mi];
        V[k * m + mi] = t;
      }
    }
  }
  for (size_t i = 0; i < m; ++i) {
    double big = 0.0, sgn = 1.0;
    for (size_t k = 0; k < m; ++k)
      if (fabs(V[k * m + i]) > big) {
        big = fabs(V[k * m + i]);
        sgn = V[k * m + i] < 0 ? -1.0 : 1.0;
      }
    for (size_t k = 0; k < m; ++k) V[k * m + i] *= sgn;
  }
}

/* Optional LAPACK basis.  The reference's ML EM depends on the signs/order LAPACK's dsyevr happens to return for the
 * d x d problem of EigenProc (see mv_jacobi above), so its `-lmm 2/4` output for d >= 3 is reproducible only with the same
 * routine.  A test can hand over the address of dsyevr_ (scipy's OpenBLAS: the one oracle/_ref/gemma is linked against);
 * EigenProc then calls it exactly as src/lapack.cpp:173-236 does (JOBZ=V, RANGE=A, UPLO=L on the row-major array, ABSTOL
 * 1e-7, eigenvector k = column k after the transpose).  Default (no hook): the Jacobi convention above. */
typedef void (*mv_dsyevr_fn)(char *, char *, char *, int *, double *, int *, double *, double *, int *, int *, double *, int *,
                             double *, double *, int *, int *, double *, int *, int *, int *, int *);
static mv_dsyevr_fn mv_dsyevr = 0;
void orc_mv_set_lapack_dsyevr(void *fn) { mv_dsyevr = (mv_dsyevr_fn)fn; }

static void mv_eig(const double *A, size_t m, double *w, double *V) {
  if (!mv_dsyevr) {
    mv_jacobi(A, m, w, V);
    return;
  }
  double a[MV_MAXD * MV_MAXD], z[MV_MAXD * MV_MAXD], work[26 * MV_MAXD + 64], vl = 0.0, vu = 0.0, abstol = 1.0e-7;
  int n = (int)m, lda = (int)m, ldz = (int)m, il = 0, iu = 0, mm = 0, info = 0, isuppz[2 * MV_MAXD];
  int lwork = (int)(sizeof(work) / sizeof(work[0])), iwork[10 * MV_MAXD + 16], liwork = 10 * MV_MAXD + 16;
  char jobz = 'V', range = 'A', uplo = 'L';
  memcpy(a, A, m * m * sizeof(double));
  mv_dsyevr(&jobz, &range, &uplo, &n, a, &lda, &vl, &vu, &il, &iu, &abstol, &mm, w, z, &ldz, isuppz, work, &lwork, iwork,
            &liwork, &info);
  for (size_t i = 0; i < m; ++i)
    for (size_t k = 0; k < m; ++k) V[i * m + k] = z[k * m + i];
}

/* LU with partial pivoting (GSL linalg/lu.c through src/lapack.cpp:307-352): inverse and log|det| */
static double mv_lu_invert(const double *A, size_t m, double *Ai) {
  double lu[MV_MAXDC * MV_MAXDC];
  size_t perm[MV_MAXDC];
  memcpy(lu, A, m * m * sizeof(double));
  for (size_t i = 0; i < m; ++i) perm[i] = i;
  for (size_t j = 0; j < m; ++j) {
    size_t ip = j;
    double amax = fabs(lu[j * m + j]);
    for (size_t i = j + 1; i < m; ++i)
      if (fabs(lu[i * m + j]) > amax) {
        amax = fabs(lu[i * m + j]);
        ip = i;
      }
    if (ip != j) {
      for (size_t k = 0; k < m; ++k) {
        const double t = lu[j * m + k];
        lu[j * m + k] = lu[ip * m + k];
        lu[ip * m + k] = t;
      }
      const size_t t = perm[j];
      perm[j] = perm[ip];
      perm[ip] = t;
    }
    const double ajj = lu[j * m + j];
    if (ajj != 0.0)
      for (size_t i = j + 1; i < m; ++i) {
        const double f = lu[i * m + j] / ajj;
        lu[i * m + j] = f;
        for (size_t k = j + 1; k < m; ++k) lu[i * m + k] -= f * lu[j * m + k];
      }
  }
  double lndet = 0.0;
  for (size_t i = 0; i < m; ++i) lndet += log(fabs(lu[i * m + i]));
  if (Ai)
    for (size_t col = 0; col < m; ++col) {
      double x[MV_MAXDC];
      for (size_t i = 0; i < m; ++i) x[i] = (perm[i] == col) ? 1.0 : 0.0;
      for (size_t i = 0; i < m; ++i)
        for (size_t k = 0; k < i; ++k) x[i] -= lu[i * m + k] * x[k];
      for (size_t ii = m; ii-- > 0;) {
        for (size_t k = ii + 1; k < m; ++k) x[ii] -= lu[ii * m + k] * x[k];
        x[ii] /= lu[ii * m + ii];
      }
      for (size_t i = 0; i < m; ++i) Ai[i * m + col] = x[i];
    }
  return lndet;
}

/* C (m x p) = op(A) * B ; ta != 0: A is k x m and used transposed, else m x k; B is k x p */
static void mv_mm(int ta, size_t m, size_t p, size_t k, const double *A, const double *B, double *C) {
  for (size_t i = 0; i < m; ++i)
    for (size_t j = 0; j < p; ++j) {
      double s = 0.0;
      for (size_t t = 0; t < k; ++t) s += (ta ? A[t * m + i] : A[i * k + t]) * B[t * p + j];
      C[i * p + j] = s;
    }
}

/* gsl_cdf_chisq_Q(x, nu) = Q(nu/2, x/2) for integer nu >= 1 (GSL cdf/chisq.c -> cdf/gamma.c): the finite sums for
 * integer and half-integer shape (all terms positive) */
double orc_cdf_chisq_Q(double x, double nu) {
  if (!(x > 0.0)) return (x == x) ? 1.0 : x;
  const int inu = (int)nu;
  const double y = 0.5 * x;
  if (inu % 2 == 0) {
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < inu / 2; ++k) {
      term *= y / k;
      sum += term;
    }
    return exp(-y) * sum;
  }
  double sum = 0.0, term = sqrt(y) / 0.886226925452758013649; /* y^(1/2) / Gamma(3/2) */
  for (int k = 1; k <= (inu - 1) / 2; ++k) {
    sum += term;
    term *= y / (k + 0.5);
  }
  return erfc(sqrt(y)) + exp(-y) * sum;
}

/* gsl_cdf_chisq_Qinv(Q, nu) = 2 gsl_cdf_gamma_Qinv(Q, nu/2, 1) (GSL cdf/chisqinv.c -> cdf/gammainv.c): a start value by range of
 * Q, then the Lagrange-corrected Newton step of gammainv.c until |step| <= 1e-10 x (at most 33 rounds).  The start value in the
 * middle range uses the normal quantile; GSL's rational approximation of it is replaced by Acklam's (relative error 1e-9)
 * polished by one Newton step on erfc -- the iteration converges to the same root to its 1e-10 tolerance either way. */
static double mv_ugaussian_Qinv(double Q) {
  static const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                              1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
  static const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                              6.680131188771972e+01, -1.328068155288572e+01};
  static const double cc[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
  static const double dd[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
  const double p = 1.0 - Q; /* lower tail */
  double x;
  if (p < 0.02425) {
    const double q = sqrt(-2.0 * log(p));
    x = (((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) / ((((dd[0] * q + dd[1]) * q + dd[2]) * q + dd[3]) * q + 1.0);
  } else if (p <= 1.0 - 0.02425) {
    const double q = p - 0.5, r = q * q;
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
  } else {
    const double q = sqrt(-2.0 * log(1.0 - p));
    x = -(((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) / ((((dd[0] * q + dd[1]) * q + dd[2]) * q + dd[3]) * q + 1.0);
  }
  const double e = 0.5 * erfc(x / sqrt(2.0)) - Q; /* Q(x) - Q; dQ/dx = -phi(x) */
  return x + e / (exp(-0.5 * x * x) / sqrt(2.0 * M_PI));
}

double orc_cdf_chisq_Qinv(double Q, double nu) {
  const double a = 0.5 * nu;
  if (Q == 1.0) return 0.0;
  if (Q == 0.0) return INFINITY;
  double x;
  if (Q < 0.05) x = -log(Q) + lgamma(a);
  else if (Q > 0.95) x = exp((lgamma(a) + log1p(-Q)) / a);
  else {
    const double xg = mv_ugaussian_Qinv(Q);
    x = (xg < -0.5 * sqrt(a)) ? a : sqrt(a) * xg + a;
  }
  for (unsigned n = 0;;) {
    const double dQ = Q - orc_cdf_chisq_Q(2.0 * x, nu);
    const double phi = exp((a - 1.0) * log(x) - x - lgamma(a)); /* gsl_ran_gamma_pdf(x, a, 1) */
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

/* PCRT (src/mvlmm.cpp:2952-2970): mode 1 Wald, 2 LRT, 3 score */
double orc_pcrt(int mode, size_t d, double p_value, double crt_a, double crt_b, double crt_c) {
  const double q = (double)d;
  const double chisq = orc_cdf_chisq_Qinv(p_value, q);
  double chisq_crt;
  if (mode == 1) {
    const double a = crt_c / (2.0 * q * (q + 2.0)), b = 1.0 + (crt_a + crt_b) / (2.0 * q);
    const double rad = b * b + 4.0 * a * chisq;
    chisq_crt = (-1.0 * b + (rad >= 0.0 ? sqrt(rad) : NAN)) / (2.0 * a); /* safe_sqrt: NaN for a negative argument */
  } else if (mode == 2) {
    chisq_crt = chisq / (1.0 + crt_a / (2.0 * q));
  } else {
    chisq_crt = chisq;
  }
  return orc_cdf_chisq_Q(chisq_crt, q);
}

/* ------------------------------------------------------------------ EigenProc, CalcQi */
/* src/mvlmm.cpp:213-282 */
static double mv_eigen_proc(size_t d, const double *Vg, const double *Ve, double *Dl, double *UltVeh, double *UltVehi) {
  double w[MV_MAXD], Ul[MV_MAXD * MV_MAXD], Veh[MV_MAXD * MV_MAXD], Vehi[MV_MAXD * MV_MAXD];
  double T1[MV_MAXD * MV_MAXD], Lam[MV_MAXD * MV_MAXD];
  double logdet_Ve = 0.0;
  mv_eig(Ve, d, w, Ul);
  memset(Veh, 0, sizeof(Veh));
  memset(Vehi, 0, sizeof(Vehi));
  for (size_t i = 0; i < d; ++i) {
    if (w[i] <= 0) continue;
    logdet_Ve += log(w[i]);
    const double s = sqrt(w[i]), si = 1.0 / s;
    for (size_t a = 0; a < d; ++a)
      for (size_t b = 0; b < d; ++b) {
        Veh[a * d + b] += s * Ul[a * d + i] * Ul[b * d + i];
        Vehi[a * d + b] += si * Ul[a * d + i] * Ul[b * d + i];
      }
  }
  mv_mm(0, d, d, d, Vg, Vehi, T1);
  mv_mm(0, d, d, d, Vehi, T1, Lam);
  if (!mv_dsyevr)
    for (size_t a = 0; a < d; ++a) /* exact symmetry for the Jacobi sweep (LAPACK reads one triangle only) */
      for (size_t b = a + 1; b < d; ++b) Lam[a * d + b] = Lam[b * d + a] = 0.5 * (Lam[a * d + b] + Lam[b * d + a]);
  mv_eig(Lam, d, Dl, Ul);
  for (size_t i = 0; i < d; ++i)
    if (Dl[i] < 0) Dl[i] = 0;
  mv_mm(1, d, d, d, Ul, Veh, UltVeh);
  mv_mm(1, d, d, d, Ul, Vehi, UltVehi);
  return logdet_Ve;
}

/* src/mvlmm.cpp:285-329: Q[(i d + l), (j d + l)] = sum_k x_ik x_jk / (D_l delta_k + 1); returns log|Q|, Qi = Q^-1 */
static double mv_calc_qi(size_t n, size_t d, size_t c, const double *eval, const double *Dl, const double *X,
                         double *Qi) {
  const size_t dc = d * c;
  double Q[MV_MAXDC * MV_MAXDC];
  memset(Q, 0, dc * dc * sizeof(double));
  for (size_t i = 0; i < c; ++i)
    for (size_t j = i; j < c; ++j)
      for (size_t l = 0; l < d; ++l) {
        double s = 0.0;
        for (size_t k = 0; k < n; ++k) s += X[i * n + k] * X[j * n + k] / (Dl[l] * eval[k] + 1.0);
        Q[(i * d + l) * dc + j * d + l] = s;
        Q[(j * d + l) * dc + i * d + l] = s;
      }
  return mv_lu_invert(Q, dc, Qi);
}

/* log|X X^T| (src/mvlmm.cpp:631-649) */
static double mv_lndet_xxt(size_t n, size_t c, const double *X, double *XXti) {
  double XXt[MV_MAXC * MV_MAXC];
  for (size_t i = 0; i < c; ++i)
    for (size_t j = i; j < c; ++j) {
      double s = 0.0;
      for (size_t k = 0; k < n; ++k) s += X[i * n + k] * X[j * n + k];
      XXt[i * c + j] = XXt[j * c + i] = s;
    }
  return mv_lu_invert(XXt, c, XXti);
}

/* ------------------------------------------------------------------ MphEM  (src/mvlmm.cpp:599-724) */
double orc_mph_em(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval,
                  const double *X, const double *Y, double *Vg, double *Ve, double *B) {
  const size_t dc = d * c;
  const int reml = (func == 'R' || func == 'r');
  double XXti[MV_MAXC * MV_MAXC];
  const double lndet_xxt = mv_lndet_xxt(n, c, X, XXti);
  const double logl_const = reml ? -0.5 * (double)(n - c) * (double)d * log(2.0 * M_PI) + 0.5 * (double)d * lndet_xxt
                                 : -0.5 * (double)n * (double)d * log(2.0 * M_PI);
  double *UltVehiY = (double *)malloc(6 * d * n * sizeof(double));
  double *UltVehiBX = UltVehiY + d * n, *UltVehiU = UltVehiBX + d * n, *UltVehiE = UltVehiU + d * n;
  double *U_hat = UltVehiE + d * n, *E_hat = U_hat + d * n;
  double Dl[MV_MAXD], UltVeh[MV_MAXD * MV_MAXD], UltVehi[MV_MAXD * MV_MAXD], UltVehiB[MV_MAXD * MV_MAXC];
  double Qi[MV_MAXDC * MV_MAXDC], xHiy[MV_MAXDC], Suu[MV_MAXD * MV_MAXD], See[MV_MAXD * MV_MAXD];
  double logl_old = 0.0, logl_new = 0.0;
  for (size_t t = 0; t < max_iter; ++t) {
    const double logdet_Ve = mv_eigen_proc(d, Vg, Ve, Dl, UltVeh, UltVehi);
    const double logdet_Q = mv_calc_qi(n, d, c, eval, Dl, X, Qi);
    mv_mm(0, d, n, d, UltVehi, Y, UltVehiY);
    /* CalcXHiY :334-359 */
    for (size_t i = 0; i < d; ++i)
      for (size_t j = 0; j < c; ++j) {
        double s = 0.0;
        for (size_t k = 0; k < n; ++k) s += X[j * n + k] * UltVehiY[i * n + k] / (eval[k] * Dl[i] + 1.0);
        xHiy[j * d + i] = s;
      }
    /* MphCalcLogL :565-594 */
    double ll = 0.0;
    for (size_t k = 0; k < n; ++k)
      for (size_t i = 0; i < d; ++i) {
        const double y = UltVehiY[i * n + k], dd = eval[k] * Dl[i] + 1.0;
        ll += y * y / dd + log(dd);
      }
    double q = 0.0;
    for (size_t a = 0; a < dc; ++a) {
      double s = 0.0;
      for (size_t b = 0; b < dc; ++b) s += Qi[a * dc + b] * xHiy[b];
      q += xHiy[a] * s;
    }
    ll -= q;
    logl_new = logl_const + (-0.5 * ll) - 0.5 * (double)n * logdet_Ve;
    if (reml) logl_new += -0.5 * (logdet_Q - (double)c * logdet_Ve);
    if (t != 0 && fabs(logl_new - logl_old) < max_prec) break;
    logl_old = logl_new;
    /* UltVehiB, UltVehiBX */
    if (reml) { /* UpdateRL_B :420-441 */
      for (size_t j = 0; j < c; ++j)
        for (size_t i = 0; i < d; ++i) {
          double s = 0.0;
          for (size_t b = 0; b < dc; ++b) s += Qi[(j * d + i) * dc + b] * xHiy[b];
          UltVehiB[i * c + j] = s;
        }
      mv_mm(0, d, n, c, UltVehiB, X, UltVehiBX);
    } else if (t == 0) {
      mv_mm(0, d, c, d, UltVehi, B, UltVehiB);
      mv_mm(0, d, n, c, UltVehiB, X, UltVehiBX);
    }
    /* UpdateU :384-391 with CalcOmega :363-382 */
    for (size_t i = 0; i < d; ++i)
      for (size_t k = 0; k < n; ++k) {
        const double ou = Dl[i] / (eval[k] * Dl[i] + 1.0), oe = eval[k] * ou;
        UltVehiU[i * n + k] = (UltVehiY[i * n + k] - UltVehiBX[i * n + k]) * oe;
      }
    if (!reml) { /* UpdateL_B :402-418 */
      double YUX[MV_MAXD * MV_MAXC];
      for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < c; ++j) {
          double s = 0.0;
          for (size_t k = 0; k < n; ++k) s += (UltVehiY[i * n + k] - UltVehiU[i * n + k]) * X[j * n + k];
          YUX[i * c + j] = s;
        }
      mv_mm(0, d, c, c, YUX, XXti, UltVehiB);
      mv_mm(0, d, n, c, UltVehiB, X, UltVehiBX);
    }
    for (size_t i = 0; i < d * n; ++i) UltVehiE[i] = UltVehiY[i] - UltVehiBX[i] - UltVehiU[i]; /* UpdateE */
    mv_mm(1, d, n, d, UltVeh, UltVehiU, U_hat);
    mv_mm(1, d, n, d, UltVeh, UltVehiE, E_hat);
    mv_mm(1, d, c, d, UltVeh, UltVehiB, B);
    /* CalcSigma :485-560 */
    memset(Suu, 0, sizeof(Suu));
    memset(See, 0, sizeof(See));
    for (size_t k = 0; k < n; ++k)
      for (size_t i = 0; i < d; ++i) {
        const double ou = Dl[i] / (eval[k] * Dl[i] + 1.0);
        Suu[i * d + i] += ou;
        See[i * d + i] += eval[k] * ou;
      }
    if (reml) {
      /* M_u[(j d + i), i] = x_j D_l_i / (delta D_l_i + 1), M_e likewise without D_l: only one entry per row, so
       * (M^T Qi M)[i1][i2] = sum_{j1 j2} m_{j1 i1} Qi[(j1 d + i1), (j2 d + i2)] m_{j2 i2} */
      for (size_t k = 0; k < n; ++k) {
        double me[MV_MAXDC], mu[MV_MAXDC];
        for (size_t i = 0; i < d; ++i)
          for (size_t j = 0; j < c; ++j) {
            const double v = X[j * n + k] / (eval[k] * Dl[i] + 1.0);
            me[j * d + i] = v;
            mu[j * d + i] = v * Dl[i];
          }
        for (size_t i1 = 0; i1 < d; ++i1)
          for (size_t i2 = 0; i2 < d; ++i2) {
            double su = 0.0, se = 0.0;
            for (size_t j1 = 0; j1 < c; ++j1)
              for (size_t j2 = 0; j2 < c; ++j2) {
                const double qv = Qi[(j1 * d + i1) * dc + j2 * d + i2];
                su += mu[j1 * d + i1] * qv * mu[j2 * d + i2];
                se += me[j1 * d + i1] * qv * me[j2 * d + i2];
              }
            Suu[i1 * d + i2] += eval[k] * su;
            See[i1 * d + i2] += se;
          }
      }
    }
    {
      double M[MV_MAXD * MV_MAXD];
      mv_mm(0, d, d, d, Suu, UltVeh, M);
      mv_mm(1, d, d, d, UltVeh, M, Suu);
      mv_mm(0, d, d, d, See, UltVeh, M);
      mv_mm(1, d, d, d, UltVeh, M, See);
    }
    /* UpdateV :443-483 */
    for (size_t a = 0; a < d; ++a)
      for (size_t b = a; b < d; ++b) {
        double sg = 0.0, se = 0.0;
        for (size_t k = 0; k < n; ++k) {
          if (eval[k] != 0) sg += U_hat[a * n + k] * U_hat[b * n + k] / eval[k];
          se += E_hat[a * n + k] * E_hat[b * n + k];
        }
        Vg[a * d + b] = Vg[b * d + a] = sg;
        Ve[a * d + b] = Ve[b * d + a] = se;
      }
    for (size_t a = 0; a < d * d; ++a) {
      Vg[a] = (Vg[a] + Suu[a]) / (double)n;
      Ve[a] = (Ve[a] + See[a]) / (double)n;
    }
  }
  free(UltVehiY);
  return logl_new;
}

/* ------------------------------------------------------------------ MphCalcP  (src/mvlmm.cpp:727-831) */
/* W: cw x n covariate rows WITHOUT the SNP, x: the SNP row.  beta[d], Vbeta[d x d]; returns the Wald p value. */
double orc_mph_calcp(size_t n, size_t d, size_t cw, const double *eval, const double *x, const double *W,
                     const double *Y, const double *Vg, const double *Ve, double *beta, double *Vbeta) {
  const size_t dc = d * cw;
  double Dl[MV_MAXD], UltVeh[MV_MAXD * MV_MAXD], UltVehi[MV_MAXD * MV_MAXD], Qi[MV_MAXDC * MV_MAXDC];
  double WHix[MV_MAXDC * MV_MAXD], QiWHix[MV_MAXDC * MV_MAXD], xPx[MV_MAXD * MV_MAXD], xPy[MV_MAXD], WHiy[MV_MAXDC];
  double *UltVehiY = (double *)malloc(d * n * sizeof(double));
  mv_eigen_proc(d, Vg, Ve, Dl, UltVeh, UltVehi);
  mv_calc_qi(n, d, cw, eval, Dl, W, Qi);
  mv_mm(0, d, n, d, UltVehi, Y, UltVehiY);
  memset(xPx, 0, sizeof(xPx));
  memset(WHix, 0, sizeof(WHix));
  for (size_t i = 0; i < d; ++i) {
    double d1 = 0.0, d2 = 0.0;
    for (size_t k = 0; k < n; ++k) {
      const double w = eval[k] * Dl[i] + 1.0;
      d1 += x[k] * UltVehiY[i * n + k] / w;
      d2 += x[k] * x[k] / w;
    }
    xPy[i] = d1;
    xPx[i * d + i] = d2;
    for (size_t j = 0; j < cw; ++j) {
      d1 = 0.0;
      d2 = 0.0;
      for (size_t k = 0; k < n; ++k) {
        const double w = eval[k] * Dl[i] + 1.0;
        d1 += x[k] * W[j * n + k] / w;
        d2 += UltVehiY[i * n + k] * W[j * n + k] / w;
      }
      WHix[(j * d + i) * d + i] = d1;
      WHiy[j * d + i] = d2;
    }
  }
  mv_mm(0, dc, d, dc, Qi, WHix, QiWHix);
  for (size_t a = 0; a < d; ++a) {
    for (size_t b = 0; b < d; ++b) {
      double s = 0.0;
      for (size_t t = 0; t < dc; ++t) s += WHix[t * d + a] * QiWHix[t * d + b];
      xPx[a * d + b] -= s;
    }
    double s = 0.0;
    for (size_t t = 0; t < dc; ++t) s += QiWHix[t * d + a] * WHiy[t];
    xPy[a] -= s;
  }
  double xPxi[MV_MAXD * MV_MAXD], sol[MV_MAXD], T[MV_MAXD * MV_MAXD];
  mv_lu_invert(xPx, d, xPxi);
  for (size_t a = 0; a < d; ++a) {
    double s = 0.0;
    for (size_t b = 0; b < d; ++b) s += xPxi[a * d + b] * xPy[b];
    sol[a] = s;
  }
  for (size_t a = 0; a < d; ++a) {
    double s = 0.0;
    for (size_t b = 0; b < d; ++b) s += UltVeh[b * d + a] * sol[b];
    beta[a] = s;
  }
  mv_mm(0, d, d, d, xPxi, UltVeh, T);
  mv_mm(1, d, d, d, UltVeh, T, Vbeta);
  double stat = 0.0;
  for (size_t a = 0; a < d; ++a) stat += sol[a] * xPy[a];
  free(UltVehiY);
  return orc_cdf_chisq_Q(stat, (double)d);
}

/* ------------------------------------------------------------------ MphNR  (src/mvlmm.cpp:2608-2760) */
/* GetIndex :1093-1109 */
static size_t mv_vindex(size_t i, size_t j, size_t d) {
  const size_t s = i < j ? i : j, l = i < j ? j : i;
  return (2 * d - s + 1) * s / 2 + l - s;
}

typedef struct {
  double logl;                       /* log (restricted) likelihood at (Vg, Ve) */
  double grad[2 * MV_MAXV];          /* d logl / d (Vg_v), then d logl / d (Ve_v) */
  double hess[4 * MV_MAXV * MV_MAXV]; /* the matrix CalcDev builds (before inversion), 2v x 2v */
  double hinv[4 * MV_MAXV * MV_MAXV]; /* its LU inverse (:2510-2518) */
  double crt[3];                     /* crt_a, crt_b, crt_c of CalcCRT at this point (c > 1; else 0) */
} mv_dev;

/* CalcCRT (src/mvlmm.cpp:2054-2331), Rothenberg's Edgeworth correction factors, in the reference's own dense form: Hinv the
 * inverse of CalcDev's Hessian (g block first), Qi (dc x dc), QM[a][v] = Qi xHiDHix (a = 1: V_g weight), M12[s][v1][v2] =
 * xHiDHiDHix (s = 0 ee, 1 ge, 2 gg; filled for v1 <= v2, which is all that is read).  Qi_s is the d x d block of the LAST
 * covariate -- the SNP. */
static double mv_sub_trace(size_t d, size_t c, const double *Mdc, const double *Qsi) { /* tr(sub(M) Qsi) */
  const size_t dc = d * c, o = (c - 1) * d;
  double t = 0.0;
  for (size_t i = 0; i < d; ++i)
    for (size_t k = 0; k < d; ++k) t += Mdc[(o + i) * dc + o + k] * Qsi[k * d + i];
  return t;
}
static void mv_sub_times(size_t d, size_t c, const double *Mdc, const double *Qsi, double *out) { /* sub(M) Qsi, d x d */
  const size_t dc = d * c, o = (c - 1) * d;
  for (size_t i = 0; i < d; ++i)
    for (size_t j = 0; j < d; ++j) {
      double t = 0.0;
      for (size_t k = 0; k < d; ++k) t += Mdc[(o + i) * dc + o + k] * Qsi[k * d + j];
      out[i * d + j] = t;
    }
}
static double mv_tr_prod(size_t d, const double *A, const double *B) {
  double t = 0.0;
  for (size_t i = 0; i < d; ++i)
    for (size_t k = 0; k < d; ++k) t += A[i * d + k] * B[k * d + i];
  return t;
}
static void mv_calc_crt(size_t d, size_t c, const double *Hinv, const double *Qi, const double *QM, const double *M12,
                        double *crt) {
  const size_t dc = d * c, m2 = dc * dc, vs = d * (d + 1) / 2, H2 = 2 * vs, o = (c - 1) * d;
  double Qs[MV_MAXD * MV_MAXD], Qsi[MV_MAXD * MV_MAXD];
  for (size_t i = 0; i < d; ++i)
    for (size_t j = 0; j < d; ++j) Qs[i * d + j] = Qi[(o + i) * dc + o + j];
  mv_lu_invert(Qs, d, Qsi);
  double *buf = (double *)malloc(11 * m2 * sizeof(double));
  double *QMQ_g1 = buf, *QMQ_e1 = buf + m2, *QMQ_g2 = buf + 2 * m2, *QMQ_e2 = buf + 3 * m2, *P_gg = buf + 4 * m2,
         *P_ge = buf + 5 * m2, *P_ee = buf + 6 * m2, *T1 = buf + 7 * m2, *T2 = buf + 8 * m2, *T3 = buf + 9 * m2,
         *T4 = buf + 10 * m2;
  double B = 0.0, Cc = 0.0, D = 0.0;
  for (size_t v1 = 0; v1 < vs; ++v1) {
    const double *QM_g1 = QM + (1 * vs + v1) * m2, *QM_e1 = QM + (0 * vs + v1) * m2;
    double A_g1[MV_MAXD * MV_MAXD], A_e1[MV_MAXD * MV_MAXD];
    mv_mm(0, dc, dc, dc, QM_g1, Qi, QMQ_g1);
    mv_mm(0, dc, dc, dc, QM_e1, Qi, QMQ_e1);
    mv_sub_times(d, c, QMQ_g1, Qsi, A_g1);
    mv_sub_times(d, c, QMQ_e1, Qsi, A_e1);
    double trCg1 = 0.0, trCe1 = 0.0;
    for (size_t k = 0; k < d; ++k) {
      trCg1 -= A_g1[k * d + k];
      trCe1 -= A_e1[k * d + k];
    }
    for (size_t v2 = v1; v2 < vs; ++v2) {
      const double *QM_g2 = QM + (1 * vs + v2) * m2, *QM_e2 = QM + (0 * vs + v2) * m2;
      double A_g2[MV_MAXD * MV_MAXD], A_e2[MV_MAXD * MV_MAXD];
      mv_mm(0, dc, dc, dc, QM_g2, Qi, QMQ_g2);
      mv_mm(0, dc, dc, dc, QM_e2, Qi, QMQ_e2);
      mv_sub_times(d, c, QMQ_g2, Qsi, A_g2);
      mv_sub_times(d, c, QMQ_e2, Qsi, A_e2);
      double trCg2 = 0.0, trCe2 = 0.0;
      for (size_t k = 0; k < d; ++k) {
        trCg2 -= A_g2[k * d + k];
        trCe2 -= A_e2[k * d + k];
      }
      const double trCC_gg = mv_tr_prod(d, A_g1, A_g2);
      const double trCC_ge = mv_tr_prod(d, A_g1, A_e2) + mv_tr_prod(d, A_e1, A_g2);
      const double trCC_ee = mv_tr_prod(d, A_e1, A_e2);
      /* Qi M Qi M Qi */
      mv_mm(0, dc, dc, dc, QM_g1, QMQ_g2, P_gg);
      mv_mm(0, dc, dc, dc, QM_g1, QMQ_e2, P_ge);
      mv_mm(0, dc, dc, dc, QM_e1, QMQ_g2, T1);
      for (size_t i = 0; i < m2; ++i) P_ge[i] += T1[i];
      mv_mm(0, dc, dc, dc, QM_e1, QMQ_e2, P_ee);
      double trB_gg = -mv_sub_trace(d, c, P_gg, Qsi), trB_ge = -mv_sub_trace(d, c, P_ge, Qsi),
             trB_ee = -mv_sub_trace(d, c, P_ee, Qsi);
      /* Qi (xHiDHiDHix) Qi */
      const double *MM_ee = M12 + ((0 * vs + v1) * vs + v2) * m2, *MM_ge = M12 + ((1 * vs + v1) * vs + v2) * m2,
                   *MM_gg = M12 + ((2 * vs + v1) * vs + v2) * m2;
      mv_mm(0, dc, dc, dc, Qi, MM_gg, T1);
      mv_mm(0, dc, dc, dc, T1, Qi, T2);
      trB_gg += mv_sub_trace(d, c, T2, Qsi);
      mv_mm(0, dc, dc, dc, Qi, MM_ge, T1);
      mv_mm(0, dc, dc, dc, T1, Qi, T3);
      trB_ge += 2.0 * mv_sub_trace(d, c, T3, Qsi);
      mv_mm(0, dc, dc, dc, Qi, MM_ee, T1);
      mv_mm(0, dc, dc, dc, T1, Qi, T4);
      trB_ee += mv_sub_trace(d, c, T4, Qsi);
      const double trD_gg = 2.0 * trB_gg, trD_ge = 2.0 * trB_ge, trD_ee = 2.0 * trB_ee;
      const double h_gg = -Hinv[v1 * H2 + v2], h_ge = -Hinv[v1 * H2 + v2 + vs], h_ee = -Hinv[(v1 + vs) * H2 + v2 + vs];
      const int times = v1 != v2 ? 2 : 1;
      for (int r = 0; r < times; ++r) {
        B += h_gg * trB_gg + h_ge * trB_ge + h_ee * trB_ee;
        Cc += h_gg * (trCC_gg + 0.5 * trCg1 * trCg2) + h_ge * (trCC_ge + 0.5 * trCg1 * trCe2 + 0.5 * trCe1 * trCg2) +
              h_ee * (trCC_ee + 0.5 * trCe1 * trCe2);
        D += h_gg * (trCC_gg + 0.5 * trD_gg) + h_ge * (trCC_ge + 0.5 * trD_ge) + h_ee * (trCC_ee + 0.5 * trD_ee);
      }
    }
  }
  free(buf);
  crt[0] = 2.0 * D - Cc;
  crt[1] = 2.0 * B;
  crt[2] = Cc;
}

/* One evaluation at (Vg, Ve): CalcHiQi :942-1012, the logl of :2697-2713 and, if want_dev, CalcDev :2360-2554.
 *
 * With G_k = H_k^-1 (d x d), b = Qi X^T H^-1 y (B = its d x c reshape) and u_k = (P y)_k = G_k (y_k - B x_k), and
 * writing the derivative direction D_v (e_i e_j^T + e_j e_i^T, or e_i e_i^T) weighted by delta_k^a (a = 1 for V_g,
 * 0 for V_e):
 *   y P D P y        = sum_k w u^T D u
 *   tr(H^-1 D)       = sum_k w tr(G D)                      tr(P D) = that - tr(Qi M_v),   M_v = sum_k w x x^T (x) G D G
 *   y P D1 P D2 P y  = sum_k w1 w2 u^T D1 G D2 u - r_v1^T Qi r_v2,                         r_v = sum_k w x (x) G D u
 *   tr(P D1 P D2)    = sum_k w1 w2 tr(G D1 G D2) - 2 tr(Qi M12) + tr(Qi M_v1 Qi M_v2),     M12 = sum_k w1 w2 x x^T (x) G D1 G D2 G
 * which are the eight-term expansions of :1863-2049 and :1526-1621 collected (P = H^-1 - H^-1 X Qi X^T H^-1).
 * The "ge" entries take D1 as the V_g direction and D2 as the V_e direction for v1 <= v2 ONLY and mirror that value
 * into the (v2, v1) slots as well (:2494-2504) -- reproduced here as is. */
static int mv_eval(int reml, size_t n, size_t d, size_t c, const double *eval, const double *X, const double *Y,
                   const double *Vg, const double *Ve, double logl_const, int want_dev, mv_dev *out) {
  const size_t dc = d * c, vs = d * (d + 1) / 2;
  double Dl[MV_MAXD], UltVeh[MV_MAXD * MV_MAXD], UltVehi[MV_MAXD * MV_MAXD], Qe[MV_MAXDC * MV_MAXDC];
  double Qi[MV_MAXDC * MV_MAXDC];
  const double logdet_Ve = mv_eigen_proc(d, Vg, Ve, Dl, UltVeh, UltVehi);
  double logdet_H = (double)n * logdet_Ve;
  double *G = (double *)malloc((d * d + d) * n * sizeof(double)); /* G_k, then Hiy_k / u_k */
  double *U = G + d * d * n;
  for (size_t k = 0; k < n; ++k) {
    double S[MV_MAXD * MV_MAXD];
    for (size_t i = 0; i < d; ++i) {
      const double dd = eval[k] * Dl[i] + 1.0;
      for (size_t j = 0; j < d; ++j) S[i * d + j] = UltVehi[i * d + j] / dd;
      logdet_H += log(dd);
    }
    mv_mm(1, d, d, d, UltVehi, S, G + k * d * d);
  }
  double logdet_Q = mv_calc_qi(n, d, c, eval, Dl, X, Qe) - (double)c * logdet_Ve;
  for (size_t i = 0; i < c; ++i) /* Qi blocks: UltVeh^T (.) UltVeh */
    for (size_t j = 0; j < c; ++j) {
      double blk[MV_MAXD * MV_MAXD], T[MV_MAXD * MV_MAXD], R[MV_MAXD * MV_MAXD];
      for (size_t a = 0; a < d; ++a)
        for (size_t b = 0; b < d; ++b) blk[a * d + b] = Qe[(i * d + a) * dc + j * d + b];
      mv_mm(0, d, d, d, blk, UltVeh, T);
      mv_mm(1, d, d, d, UltVeh, T, R);
      for (size_t a = 0; a < d; ++a)
        for (size_t b = 0; b < d; ++b) Qi[(i * d + a) * dc + j * d + b] = R[a * d + b];
    }
  /* Hiy, xHiy, b = Qi xHiy, yPy */
  double xHiy[MV_MAXDC], b[MV_MAXDC], yHiy = 0.0;
  memset(xHiy, 0, sizeof(xHiy));
  for (size_t k = 0; k < n; ++k) {
    const double *Gk = G + k * d * d;
    for (size_t i = 0; i < d; ++i) {
      double s = 0.0;
      for (size_t j = 0; j < d; ++j) s += Gk[i * d + j] * Y[j * n + k];
      U[k * d + i] = s;
      yHiy += s * Y[i * n + k];
      for (size_t j = 0; j < c; ++j) xHiy[j * d + i] += X[j * n + k] * s;
    }
  }
  double yPy = yHiy;
  for (size_t a = 0; a < dc; ++a) {
    double s = 0.0;
    for (size_t t = 0; t < dc; ++t) s += Qi[a * dc + t] * xHiy[t];
    b[a] = s;
    yPy -= s * xHiy[a];
  }
  out->logl = reml ? logl_const - 0.5 * logdet_H - 0.5 * logdet_Q - 0.5 * yPy : logl_const - 0.5 * logdet_H - 0.5 * yPy;
  if (!want_dev) {
    free(G);
    return 0;
  }
  /* u_k = Hiy_k - G_k B x_k */
  for (size_t k = 0; k < n; ++k) {
    const double *Gk = G + k * d * d;
    double bx[MV_MAXD];
    for (size_t i = 0; i < d; ++i) {
      double s = 0.0;
      for (size_t j = 0; j < c; ++j) s += b[j * d + i] * X[j * n + k];
      bx[i] = s;
    }
    for (size_t i = 0; i < d; ++i) {
      double s = 0.0;
      for (size_t j = 0; j < d; ++j) s += Gk[i * d + j] * bx[j];
      U[k * d + i] -= s;
    }
  }
  /* direction matrices */
  double Dm[MV_MAXV][MV_MAXD * MV_MAXD];
  for (size_t i = 0; i < d; ++i)
    for (size_t j = i; j < d; ++j) {
      double *D = Dm[mv_vindex(i, j, d)];
      memset(D, 0, d * d * sizeof(double));
      D[i * d + j] = 1.0;
      D[j * d + i] = 1.0;
    }
  /* accumulators: index a = 0 (V_e weight 1), 1 (V_g weight delta); pair weight index s = 0 (ee), 1 (ge), 2 (gg) */
  const size_t m2 = dc * dc;
  double *acc = (double *)calloc(2 * vs * (2 + dc + m2) + 3 * vs * vs * (2 + m2), sizeof(double));
  double *yPDPy = acc, *trHiD = yPDPy + 2 * vs, *rv = trHiD + 2 * vs, *Mv = rv + 2 * vs * dc;
  double *yy = Mv + 2 * vs * m2, *trHH = yy + 3 * vs * vs, *M12 = trHH + 3 * vs * vs;
  for (size_t k = 0; k < n; ++k) {
    const double *Gk = G + k * d * d, *u = U + k * d;
    const double dl = eval[k];
    double t[MV_MAXV][MV_MAXD], Gt[MV_MAXV][MV_MAXD], GDG[MV_MAXV][MV_MAXD * MV_MAXD], GD[MV_MAXD * MV_MAXD];
    for (size_t v = 0; v < vs; ++v) {
      for (size_t i = 0; i < d; ++i) {
        double s = 0.0;
        for (size_t j = 0; j < d; ++j) s += Dm[v][i * d + j] * u[j];
        t[v][i] = s;
      }
      for (size_t i = 0; i < d; ++i) {
        double s = 0.0;
        for (size_t j = 0; j < d; ++j) s += Gk[i * d + j] * t[v][j];
        Gt[v][i] = s;
      }
      mv_mm(0, d, d, d, Gk, Dm[v], GD);
      mv_mm(0, d, d, d, GD, Gk, GDG[v]);
      double udu = 0.0, tr = 0.0;
      for (size_t i = 0; i < d; ++i) {
        udu += u[i] * t[v][i];
        tr += GD[i * d + i];
      }
      for (int a = 0; a < 2; ++a) {
        const double w = a ? dl : 1.0;
        yPDPy[a * vs + v] += w * udu;
        trHiD[a * vs + v] += w * tr;
        for (size_t j = 0; j < c; ++j)
          for (size_t i = 0; i < d; ++i) rv[(a * vs + v) * dc + j * d + i] += w * X[j * n + k] * Gt[v][i];
        for (size_t j1 = 0; j1 < c; ++j1)
          for (size_t j2 = 0; j2 < c; ++j2) {
            const double xx = w * X[j1 * n + k] * X[j2 * n + k];
            for (size_t i1 = 0; i1 < d; ++i1)
              for (size_t i2 = 0; i2 < d; ++i2)
                Mv[(a * vs + v) * m2 + (j1 * d + i1) * dc + j2 * d + i2] += xx * GDG[v][i1 * d + i2];
          }
      }
    }
    for (size_t v1 = 0; v1 < vs; ++v1)
      for (size_t v2 = v1; v2 < vs; ++v2) {
        double s = 0.0, tr = 0.0, GDGD[MV_MAXD * MV_MAXD], GDGDG[MV_MAXD * MV_MAXD];
        for (size_t i = 0; i < d; ++i) s += t[v1][i] * Gt[v2][i];
        mv_mm(0, d, d, d, GDG[v1], Dm[v2], GDGD);
        for (size_t i = 0; i < d; ++i) tr += GDGD[i * d + i];
        mv_mm(0, d, d, d, GDGD, Gk, GDGDG);
        for (int sI = 0; sI < 3; ++sI) {
          const double w = sI == 0 ? 1.0 : (sI == 1 ? dl : dl * dl);
          const size_t p = (sI * vs + v1) * vs + v2;
          yy[p] += w * s;
          trHH[p] += w * tr;
          for (size_t j1 = 0; j1 < c; ++j1)
            for (size_t j2 = 0; j2 < c; ++j2) {
              const double xx = w * X[j1 * n + k] * X[j2 * n + k];
              for (size_t i1 = 0; i1 < d; ++i1)
                for (size_t i2 = 0; i2 < d; ++i2)
                  M12[p * m2 + (j1 * d + i1) * dc + j2 * d + i2] += xx * GDGDG[i1 * d + i2];
            }
        }
      }
  }
  /* Qi M_v, Qi r_v */
  double *QM = (double *)malloc(2 * vs * (m2 + dc) * sizeof(double)), *Qr = QM + 2 * vs * m2;
  for (size_t av = 0; av < 2 * vs; ++av) {
    mv_mm(0, dc, dc, dc, Qi, Mv + av * m2, QM + av * m2);
    for (size_t a = 0; a < dc; ++a) {
      double s = 0.0;
      for (size_t t2 = 0; t2 < dc; ++t2) s += Qi[a * dc + t2] * rv[av * dc + t2];
      Qr[av * dc + a] = s;
    }
  }
  const size_t H2 = 2 * vs;
  memset(out->hess, 0, sizeof(out->hess));
  for (size_t v1 = 0; v1 < vs; ++v1) {
    for (int a = 0; a < 2; ++a) {
      double trPD = trHiD[a * vs + v1];
      if (reml)
        for (size_t i = 0; i < dc; ++i) trPD -= QM[(a * vs + v1) * m2 + i * dc + i];
      out->grad[(a ? 0 : vs) + v1] = -0.5 * trPD + 0.5 * yPDPy[a * vs + v1];
    }
    for (size_t v2 = v1; v2 < vs; ++v2) {
      double dev2[3]; /* ee, ge, gg */
      for (int sI = 0; sI < 3; ++sI) {
        const int a1 = sI >= 1, a2 = sI == 2; /* ge: D1 = g (v1), D2 = e (v2) */
        const size_t p = (sI * vs + v1) * vs + v2;
        double yPDPDPy = yy[p];
        for (size_t i = 0; i < dc; ++i) yPDPDPy -= rv[(a1 * vs + v1) * dc + i] * Qr[(a2 * vs + v2) * dc + i];
        double tr = trHH[p];
        if (reml) {
          double t2 = 0.0, t4 = 0.0;
          for (size_t i = 0; i < dc; ++i)
            for (size_t j = 0; j < dc; ++j) {
              t2 += Qi[i * dc + j] * M12[p * m2 + j * dc + i];
              t4 += QM[(a1 * vs + v1) * m2 + i * dc + j] * QM[(a2 * vs + v2) * m2 + j * dc + i];
            }
          tr += -2.0 * t2 + t4;
        }
        dev2[sI] = 0.5 * tr - yPDPDPy;
      }
      out->hess[v1 * H2 + v2] = out->hess[v2 * H2 + v1] = dev2[2];
      out->hess[(v1 + vs) * H2 + v2 + vs] = out->hess[(v2 + vs) * H2 + v1 + vs] = dev2[0];
      out->hess[v1 * H2 + v2 + vs] = out->hess[(v2 + vs) * H2 + v1] = dev2[1];
      out->hess[v2 * H2 + v1 + vs] = out->hess[(v1 + vs) * H2 + v2] = dev2[1];
    }
  }
  mv_lu_invert(out->hess, H2, out->hinv);
  out->crt[0] = out->crt[1] = out->crt[2] = 0.0;
  if (c > 1) mv_calc_crt(d, c, out->hinv, Qi, QM, M12, out->crt); /* :2522-2530: for 'R' and 'L' alike */
  free(QM);
  free(acc);
  free(G);
  return 0;
}

static int mv_is_pd(size_t d, const double *V) {
  double w[MV_MAXD], Z[MV_MAXD * MV_MAXD];
  mv_jacobi(V, d, w, Z);
  for (size_t i = 0; i < d; ++i)
    if (w[i] <= 0) return 0;
  return 1;
}

/* Hessian_inv receives MINUS the inverse of the last Hessian (the variance matrix, :2742-2744); 2v x 2v */
double orc_mph_nr_crt(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval,
                      const double *X, const double *Y, double *Vg, double *Ve, double *Hessian_inv, double *crt /* 3 or NULL */) {
  const int reml = (func == 'R' || func == 'r');
  const size_t vs = d * (d + 1) / 2, H2 = 2 * vs;
  const double lndet_xxt = mv_lndet_xxt(n, c, X, NULL);
  const double logl_const = reml ? -0.5 * (double)(n - c) * (double)d * log(2.0 * M_PI) + 0.5 * (double)d * lndet_xxt
                                 : -0.5 * (double)n * (double)d * log(2.0 * M_PI);
  mv_dev *ev = (mv_dev *)malloc(sizeof(mv_dev));
  double Vg_save[MV_MAXD * MV_MAXD], Ve_save[MV_MAXD * MV_MAXD], Hi[4 * MV_MAXV * MV_MAXV], grad[2 * MV_MAXV];
  double logl_old = 0.0, logl_new = 0.0, crt_last[3] = {0.0, 0.0, 0.0};
  memset(Hi, 0, sizeof(Hi));
  memset(grad, 0, sizeof(grad));
  for (size_t t = 0; t < max_iter; ++t) {
    memcpy(Vg_save, Vg, d * d * sizeof(double));
    memcpy(Ve_save, Ve, d * d * sizeof(double));
    double step_scale = 1.0;
    size_t step_iter = 0;
    int flag_pd;
    do {
      memcpy(Vg, Vg_save, d * d * sizeof(double));
      memcpy(Ve, Ve_save, d * d * sizeof(double));
      if (t != 0) { /* UpdateVgVe :2557-2606 */
        for (size_t i = 0; i < d; ++i)
          for (size_t j = i; j < d; ++j) {
            const size_t v = mv_vindex(i, j, d);
            double sg = 0.0, se = 0.0;
            for (size_t q = 0; q < H2; ++q) {
              sg += Hi[v * H2 + q] * grad[q];
              se += Hi[(v + vs) * H2 + q] * grad[q];
            }
            Vg[i * d + j] = Vg[j * d + i] = Vg_save[i * d + j] - step_scale * sg;
            Ve[i * d + j] = Ve[j * d + i] = Ve_save[i * d + j] - step_scale * se;
          }
      }
      flag_pd = mv_is_pd(d, Ve) && mv_is_pd(d, Vg);
      if (flag_pd) {
        mv_eval(reml, n, d, c, eval, X, Y, Vg, Ve, logl_const, 0, ev);
        logl_new = ev->logl;
      }
      step_scale /= 2.0;
      step_iter++;
    } while ((flag_pd == 0 || logl_new < logl_old || logl_new - logl_old > 10) && step_iter < 10 && t != 0);
    if (t != 0) {
      if (logl_new < logl_old || flag_pd == 0) {
        memcpy(Vg, Vg_save, d * d * sizeof(double));
        memcpy(Ve, Ve_save, d * d * sizeof(double));
        break;
      }
      if (logl_new - logl_old < max_prec) break;
    }
    logl_old = logl_new;
    mv_eval(reml, n, d, c, eval, X, Y, Vg, Ve, logl_const, 1, ev);
    memcpy(grad, ev->grad, H2 * sizeof(double));
    memcpy(Hi, ev->hinv, H2 * H2 * sizeof(double));
    memcpy(crt_last, ev->crt, sizeof(crt_last)); /* the factors of the LAST CalcDev call are what MphNR hands back */
  }
  if (Hessian_inv)
    for (size_t i = 0; i < H2 * H2; ++i) Hessian_inv[i] = -Hi[i];
  if (crt) memcpy(crt, crt_last, sizeof(crt_last));
  free(ev);
  return logl_new;
}
double orc_mph_nr(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval,
                  const double *X, const double *Y, double *Vg, double *Ve, double *Hessian_inv) {
  return orc_mph_nr_crt(func, max_iter, max_prec, n, d, c, eval, X, Y, Vg, Ve, Hessian_inv, NULL);
}

/* test hook: logl, gradient and the CalcDev Hessian at a point (finite-difference checks) */
double orc_mph_dev(char func, size_t n, size_t d, size_t c, const double *eval, const double *X, const double *Y,
                   const double *Vg, const double *Ve, double *grad, double *hess) {
  const int reml = (func == 'R' || func == 'r');
  const size_t H2 = d * (d + 1);
  const double lndet_xxt = mv_lndet_xxt(n, c, X, NULL);
  const double logl_const = reml ? -0.5 * (double)(n - c) * (double)d * log(2.0 * M_PI) + 0.5 * (double)d * lndet_xxt
                                 : -0.5 * (double)n * (double)d * log(2.0 * M_PI);
  mv_dev *ev = (mv_dev *)malloc(sizeof(mv_dev));
  mv_eval(reml, n, d, c, eval, X, Y, Vg, Ve, logl_const, grad != NULL, ev);
  const double ll = ev->logl;
  if (grad) {
    memcpy(grad, ev->grad, H2 * sizeof(double));
    memcpy(hess, ev->hess, H2 * H2 * sizeof(double));
  }
  free(ev);
  return ll;
}

/* B = GLS estimate of the fixed effects at (Vg, Ve): the tail of MphInitial (:2886-2935) and the B that MphCalcBeta
 * (:835-935) stores (its standard errors are not needed on this path) */
static void mv_gls_B(size_t n, size_t d, size_t c, const double *eval, const double *X, const double *Y,
                     const double *Vg, const double *Ve, double *B) {
  double Dl[MV_MAXD], UltVeh[MV_MAXD * MV_MAXD], UltVehi[MV_MAXD * MV_MAXD], Qi[MV_MAXDC * MV_MAXDC];
  double xHiy[MV_MAXDC], beta[MV_MAXDC];
  const size_t dc = d * c;
  double *UltVehiY = (double *)malloc(d * n * sizeof(double));
  mv_eigen_proc(d, Vg, Ve, Dl, UltVeh, UltVehi);
  mv_calc_qi(n, d, c, eval, Dl, X, Qi);
  mv_mm(0, d, n, d, UltVehi, Y, UltVehiY);
  for (size_t i = 0; i < d; ++i)
    for (size_t j = 0; j < c; ++j) {
      double s = 0.0;
      for (size_t k = 0; k < n; ++k) s += UltVehiY[i * n + k] * X[j * n + k] / (eval[k] * Dl[i] + 1.0);
      xHiy[j * d + i] = s;
    }
  for (size_t a = 0; a < dc; ++a) {
    double s = 0.0;
    for (size_t t = 0; t < dc; ++t) s += Qi[a * dc + t] * xHiy[t];
    beta[a] = s;
  }
  for (size_t j = 0; j < c; ++j)
    for (size_t a = 0; a < d; ++a) {
      double s = 0.0;
      for (size_t t = 0; t < d; ++t) s += UltVeh[t * d + a] * beta[j * d + t];
      B[a * c + j] = s;
    }
  free(UltVehiY);
}

/* ------------------------------------------------------------------ MphInitial  (src/mvlmm.cpp:2763-2948) */
void orc_mph_initial(size_t em_iter, double em_prec, size_t nr_iter, double nr_prec, size_t n, size_t d, size_t c,
                     const double *eval, const double *X, const double *Y, double l_min, double l_max, size_t n_region,
                     double *Vg, double *Ve, double *B) {
  memset(Vg, 0, d * d * sizeof(double));
  memset(Ve, 0, d * d * sizeof(double));
  memset(B, 0, d * c * sizeof(double));
  double *Xt = (double *)malloc(n * c * sizeof(double));
  for (size_t k = 0; k < n; ++k)
    for (size_t j = 0; j < c; ++j) Xt[k * c + j] = X[j * n + k];
  for (size_t i = 0; i < d; ++i) {
    double lambda, logl, vg, ve, bt[MV_MAXC], sb[MV_MAXC];
    orc_CalcLambda_null('R', n, c, eval, Xt, Y + i * n, l_min, l_max, n_region, &lambda, &logl);
    orc_CalcLmmVgVeBeta(n, c, eval, Xt, Y + i * n, lambda, &vg, &ve, bt, sb);
    Vg[i * d + i] = vg;
    Ve[i * d + i] = ve;
  }
  free(Xt);
  if (d > 4) { /* pairwise two-trait fits for the off-diagonals */
    double *Ys = (double *)malloc(2 * n * sizeof(double));
    for (size_t i = 0; i < d; ++i) {
      memcpy(Ys, Y + i * n, n * sizeof(double));
      for (size_t j = i + 1; j < d; ++j) {
        memcpy(Ys + n, Y + j * n, n * sizeof(double));
        double Vgs[4] = {Vg[i * d + i], 0, 0, Vg[j * d + j]}, Ves[4] = {Ve[i * d + i], 0, 0, Ve[j * d + j]};
        double Bs[2 * MV_MAXC];
        orc_mph_em('R', em_iter, em_prec, n, 2, c, eval, X, Ys, Vgs, Ves, Bs);
        orc_mph_nr('R', nr_iter, nr_prec, n, 2, c, eval, X, Ys, Vgs, Ves, NULL);
        Vg[i * d + j] = Vg[j * d + i] = Vgs[1];
        Ve[i * d + j] = Ve[j * d + i] = Ves[1];
      }
    }
    free(Ys);
  }
  mv_gls_B(n, d, c, eval, X, Y, Vg, Ve, B);
}

/* ------------------------------------------------------------------ the drivers */
typedef struct {
  size_t em_iter, nr_iter, n_region;
  double em_prec, nr_prec, l_min, l_max, p_nr;
  size_t crt; /* -crt: Edgeworth-corrected p values (PCRT) for the SNPs that reach the Newton-Raphson stage */
} orc_mv_cfg;

/* The null-model block of MVLMM::AnalyzeBimbam, src/mvlmm.cpp:3056-3208.  W: cw x n.  Outputs the REMLE and MLE fits
 * (Vg, Ve, B as d x cw), both log-likelihoods; the per-SNP loop starts from the MLE fit (:3206-3208). */
void orc_mvlmm_null(const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval, const double *W,
                    const double *Y, double *Vg_remle, double *Ve_remle, double *B_remle, double *logl_remle,
                    double *Vg_mle, double *Ve_mle, double *B_mle, double *logl_mle) {
  double Vg[MV_MAXD * MV_MAXD], Ve[MV_MAXD * MV_MAXD], B[MV_MAXD * MV_MAXC];
  orc_mph_initial(cfg->em_iter, cfg->em_prec, cfg->nr_iter, cfg->nr_prec, n, d, cw, eval, W, Y, cfg->l_min, cfg->l_max,
                  cfg->n_region, Vg, Ve, B);
  orc_mph_em('R', cfg->em_iter, cfg->em_prec, n, d, cw, eval, W, Y, Vg, Ve, B);
  *logl_remle = orc_mph_nr('R', cfg->nr_iter, cfg->nr_prec, n, d, cw, eval, W, Y, Vg, Ve, NULL);
  mv_gls_B(n, d, cw, eval, W, Y, Vg, Ve, B); /* MphCalcBeta :3070 */
  memcpy(Vg_remle, Vg, d * d * sizeof(double));
  memcpy(Ve_remle, Ve, d * d * sizeof(double));
  memcpy(B_remle, B, d * cw * sizeof(double));
  orc_mph_em('L', cfg->em_iter, cfg->em_prec, n, d, cw, eval, W, Y, Vg, Ve, B);
  *logl_mle = orc_mph_nr('L', cfg->nr_iter, cfg->nr_prec, n, d, cw, eval, W, Y, Vg, Ve, NULL);
  mv_gls_B(n, d, cw, eval, W, Y, Vg, Ve, B); /* MphCalcBeta :3137 */
  memcpy(Vg_mle, Vg, d * d * sizeof(double));
  memcpy(Ve_mle, Ve, d * d * sizeof(double));
  memcpy(B_mle, B, d * cw * sizeof(double));
}

/* The per-SNP block, src/mvlmm.cpp:3287-3374.  UtX: l x n (SNP-major).  out per SNP: beta[d], Vbeta[v], Vg[v], Ve[v],
 * p_wald, p_lrt, p_score  (stride 3 v + d + 3).  B_null is d x cw. */
void orc_mvlmm_batch(int a_mode, const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval,
                     const double *W, const double *Y, const double *UtX, size_t l, const double *Vg_null,
                     const double *Ve_null, const double *B_null, double logl_H0, double *out) {
  const size_t c = cw + 1, vs = d * (d + 1) / 2, stride = 3 * vs + d + 3;
  double *X = (double *)malloc(c * n * sizeof(double));
  memcpy(X, W, cw * n * sizeof(double));
  for (size_t s = 0; s < l; ++s) {
    const double *x = UtX + s * n;
    memcpy(X + cw * n, x, n * sizeof(double));
    double Vg[MV_MAXD * MV_MAXD], Ve[MV_MAXD * MV_MAXD], B[MV_MAXD * MV_MAXC], beta[MV_MAXD], Vbeta[MV_MAXD * MV_MAXD];
    double p_wald = 0, p_lrt = 0, p_score = 0, logl_H1;
    memset(beta, 0, sizeof(beta));
    memset(Vbeta, 0, sizeof(Vbeta));
    memcpy(Vg, Vg_null, d * d * sizeof(double));
    memcpy(Ve, Ve_null, d * d * sizeof(double));
    for (size_t i = 0; i < d; ++i) {
      for (size_t j = 0; j < cw; ++j) B[i * c + j] = B_null[i * cw + j];
      B[i * c + cw] = 0.0;
    }
    double crt[3] = {0.0, 0.0, 0.0};
    if (a_mode == 3 || a_mode == 4) {
      p_score = orc_mph_calcp(n, d, cw, eval, x, W, Y, Vg_null, Ve_null, beta, Vbeta);
      if (p_score < cfg->p_nr && cfg->crt == 1) { /* :3302-3306: one CalcDev at the null estimates, PCRT mode 3 */
        logl_H1 = orc_mph_nr_crt('R', 1, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        p_score = orc_pcrt(3, d, p_score, crt[0], crt[1], crt[2]);
      }
    }
    if (a_mode == 2 || a_mode == 4) {
      logl_H1 = orc_mph_em('L', cfg->em_iter / 10, cfg->em_prec * 10, n, d, c, eval, X, Y, Vg, Ve, B);
      orc_mph_calcp(n, d, cw, eval, x, W, Y, Vg, Ve, beta, Vbeta);
      p_lrt = orc_cdf_chisq_Q(2.0 * (logl_H1 - logl_H0), (double)d);
      if (p_lrt < cfg->p_nr) {
        logl_H1 = orc_mph_nr_crt('L', cfg->nr_iter / 10, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        orc_mph_calcp(n, d, cw, eval, x, W, Y, Vg, Ve, beta, Vbeta);
        p_lrt = orc_cdf_chisq_Q(2.0 * (logl_H1 - logl_H0), (double)d);
        if (cfg->crt == 1) p_lrt = orc_pcrt(2, d, p_lrt, crt[0], crt[1], crt[2]);
      }
    }
    if (a_mode == 1 || a_mode == 4) {
      orc_mph_em('R', cfg->em_iter / 10, cfg->em_prec * 10, n, d, c, eval, X, Y, Vg, Ve, B);
      p_wald = orc_mph_calcp(n, d, cw, eval, x, W, Y, Vg, Ve, beta, Vbeta);
      if (p_wald < cfg->p_nr) {
        orc_mph_nr_crt('R', cfg->nr_iter / 10, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        p_wald = orc_mph_calcp(n, d, cw, eval, x, W, Y, Vg, Ve, beta, Vbeta);
        if (cfg->crt == 1) p_wald = orc_pcrt(1, d, p_wald, crt[0], crt[1], crt[2]);
      }
    }
    double *o = out + s * stride;
    for (size_t i = 0; i < d; ++i) o[i] = beta[i];
    size_t q = 0;
    for (size_t i = 0; i < d; ++i)
      for (size_t j = i; j < d; ++j, ++q) {
        o[d + q] = Vbeta[i * d + j];
        o[d + vs + q] = Vg[i * d + j];
        o[d + 2 * vs + q] = Ve[i * d + j];
      }
    o[d + 3 * vs] = p_wald;
    o[d + 3 * vs + 1] = p_lrt;
    o[d + 3 * vs + 2] = p_score;
  }
  free(X);
}

/* The per-SNP block of MVLMM::AnalyzeBimbamGXE, src/mvlmm.cpp:4253-4348 (AnalyzePlinkGXE :4700-4795 is the same).  W: cw x n, its
 * last row U^T env (X_sub1 of the reference); UtX, UtX2: l x n, the rotated SNP and its product with env (X_row1, X_row2).
 * Per SNP the null of the test, X_sub2 = (W, x), is fitted first -- REML for modes 3 / 4 (:4262-4272), then ML for modes 2 / 4
 * (:4274-4284) -- with V_g, V_e and B carried from one fit to the next exactly as the reference's variables are; the tested row
 * is x o env with X_sub2 as covariates.  B_null: d x cw.  The allele switch of :4232-4236 (and beta's sign, :4331-4333) is the
 * caller's.  out as orc_mvlmm_batch. */
void orc_mvlmm_batch_gxe(int a_mode, const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval, const double *W,
                         const double *Y, const double *UtX, const double *UtX2, size_t l, const double *Vg_null,
                         const double *Ve_null, const double *B_null, double *out) {
  const size_t c2 = cw + 1, c = cw + 2, vs = d * (d + 1) / 2, stride = 3 * vs + d + 3; /* rows of X_sub2, of X */
  double *X = (double *)malloc(c * n * sizeof(double));
  memcpy(X, W, cw * n * sizeof(double));
  for (size_t s = 0; s < l; ++s) {
    const double *x = UtX + s * n, *x2 = UtX2 + s * n;
    memcpy(X + cw * n, x, n * sizeof(double));
    memcpy(X + c2 * n, x2, n * sizeof(double));
    double Vg[MV_MAXD * MV_MAXD], Ve[MV_MAXD * MV_MAXD], B[MV_MAXD * MV_MAXC], B2[MV_MAXD * MV_MAXC], beta[MV_MAXD], Vbeta[MV_MAXD * MV_MAXD];
    double p_wald = 0, p_lrt = 0, p_score = 0, logl_H1, logl_H0 = 0.0;
    memset(beta, 0, sizeof(beta));
    memset(Vbeta, 0, sizeof(Vbeta));
    memcpy(Vg, Vg_null, d * d * sizeof(double));
    memcpy(Ve, Ve_null, d * d * sizeof(double));
    for (size_t i = 0; i < d; ++i) { /* B = B_null: the global null's columns, zeros for the SNP and the interaction (:4056-4059) */
      for (size_t j = 0; j < cw; ++j) B[i * c + j] = B_null[i * cw + j];
      B[i * c + cw] = B[i * c + cw + 1] = 0.0;
    }
    double crt[3] = {0.0, 0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) { /* the per-SNP null on X_sub2: 'R' then 'L' */
      const char f = pass == 0 ? 'R' : 'L';
      if (pass == 0 && !(a_mode == 3 || a_mode == 4)) continue;
      if (pass == 1 && !(a_mode == 2 || a_mode == 4)) continue;
      for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < c2; ++j) B2[i * c2 + j] = B[i * c + j]; /* B_sub2: a view of B's first c2 columns */
      logl_H0 = orc_mph_em(f, cfg->em_iter / 10, cfg->em_prec * 10, n, d, c2, eval, X, Y, Vg, Ve, B2);
      logl_H0 = orc_mph_nr(f, cfg->nr_iter / 10, cfg->nr_prec * 10, n, d, c2, eval, X, Y, Vg, Ve, NULL);
      mv_gls_B(n, d, c2, eval, X, Y, Vg, Ve, B2); /* MphCalcBeta */
      for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < c2; ++j) B[i * c + j] = B2[i * c2 + j];
    }
    if (a_mode == 3 || a_mode == 4) {
      p_score = orc_mph_calcp(n, d, c2, eval, x2, X, Y, Vg_null, Ve_null, beta, Vbeta);
      if (p_score < cfg->p_nr && cfg->crt == 1) {
        logl_H1 = orc_mph_nr_crt('R', 1, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        p_score = orc_pcrt(3, d, p_score, crt[0], crt[1], crt[2]);
      }
    }
    if (a_mode == 2 || a_mode == 4) {
      logl_H1 = orc_mph_em('L', cfg->em_iter / 10, cfg->em_prec * 10, n, d, c, eval, X, Y, Vg, Ve, B);
      orc_mph_calcp(n, d, c2, eval, x2, X, Y, Vg, Ve, beta, Vbeta);
      p_lrt = orc_cdf_chisq_Q(2.0 * (logl_H1 - logl_H0), (double)d);
      if (p_lrt < cfg->p_nr) {
        logl_H1 = orc_mph_nr_crt('L', cfg->nr_iter / 10, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        orc_mph_calcp(n, d, c2, eval, x2, X, Y, Vg, Ve, beta, Vbeta);
        p_lrt = orc_cdf_chisq_Q(2.0 * (logl_H1 - logl_H0), (double)d);
        if (cfg->crt == 1) p_lrt = orc_pcrt(2, d, p_lrt, crt[0], crt[1], crt[2]);
      }
    }
    if (a_mode == 1 || a_mode == 4) {
      orc_mph_em('R', cfg->em_iter / 10, cfg->em_prec * 10, n, d, c, eval, X, Y, Vg, Ve, B);
      p_wald = orc_mph_calcp(n, d, c2, eval, x2, X, Y, Vg, Ve, beta, Vbeta);
      if (p_wald < cfg->p_nr) {
        orc_mph_nr_crt('R', cfg->nr_iter / 10, cfg->nr_prec * 10, n, d, c, eval, X, Y, Vg, Ve, NULL, crt);
        p_wald = orc_mph_calcp(n, d, c2, eval, x2, X, Y, Vg, Ve, beta, Vbeta);
        if (cfg->crt == 1) p_wald = orc_pcrt(1, d, p_wald, crt[0], crt[1], crt[2]);
      }
    }
    double *o = out + s * stride;
    for (size_t i = 0; i < d; ++i) o[i] = beta[i];
    size_t q = 0;
    for (size_t i = 0; i < d; ++i)
      for (size_t j = i; j < d; ++j, ++q) {
        o[d + q] = Vbeta[i * d + j];
        o[d + vs + q] = Vg[i * d + j];
        o[d + 2 * vs + q] = Ve[i * d + j];
      }
    o[d + 3 * vs] = p_wald;
    o[d + 3 * vs + 1] = p_lrt;
    o[d + 3 * vs + 2] = p_score;
  }
  free(X);
}
