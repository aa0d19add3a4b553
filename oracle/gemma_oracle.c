/*
 * gemma_oracle.c -- CPU restatement of GEMMA's kinship + univariate-LMM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gemma_amd/ may include, link,
 * dlopen or call this file; it exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg have an independent checker for the HIP path.
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * restates.  The reference's arithmetic lives partly in GSL 2.x (not vendored
 * in the reference tree, "GSL 2.x" INSTALL.md:17): the Brent / Newton root
 * finders, gsl_root_test_interval/delta and gsl_cdf_fdist_Q / gsl_cdf_chisq_Q
 * are restated here from GSL's published algorithms (roots/brent.c,
 * roots/newton.c, roots/convergence.c, cdf/fdist.c, cdf/beta_inc.c,
 * cdf/gamma.c).  PINNING, two layers: (1) the reference's own golden values
 * for BXD (test/dev_tests.rb:26-55, test/dev_test_suite.sh:51-52), see
 * tests/test_oracle_golden.py; (2) outputs of the reference itself -- its
 * sources compiled unchanged into oracle/_ref/gemma (oracle/Makefile `ref`,
 * GSL API from oracle/gslshim) and run on BXD, issue188 and issue243 -- which
 * this file reproduces to every printed digit for -gk 1/2, -lmm 1/2/3/4/9,
 * covariates, -lm 1..4 and GXE (tests/test_reference_pin.py, fixtures
 * tests/golden/ref_*.npz).
 *
 * Plain C99, links libm only.  Dense linear algebra that the reference hands
 * to OpenBLAS (cblas_dgemm, dsyevr_) is done by the Python side of the oracle
 * (oracle/oracle.py) through numpy/scipy's bundled OpenBLAS, i.e. the same
 * routines the reference calls.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_P_YY_MIN 1e-8 /* src/lmm.cpp:52 */

typedef struct {
  double beta, se, lambda_remle, lambda_mle, p_wald, p_lrt, p_score, logl_H1;
} orc_sumstat; /* == SUMSTAT, src/param.h:54-66 */

/* ------------------------------------------------------------------ */
/* src/param.cpp:1400-1415  GetabIndex (1-based a,b; upper triangle)   */
size_t orc_GetabIndex(size_t a, size_t b, size_t n_cvt) {
  size_t cols = n_cvt + 2;
  size_t a1 = a, b1 = b;
  if (b <= a) {
    a1 = b;
    b1 = a;
  }
  return (2 * cols - a1 + 2) * (a1 - 1) / 2 + b1 - a1;
}

/* src/mathfunc.cpp:122-131  safe_sqrt: note `fabs(d < 0.001)` in the
 * reference is fabs() of a boolean, i.e. the test is simply d < 0.001. */
double orc_safe_sqrt(double d) {
  double d1 = d;
  if (d < 0.001) d1 = fabs(d);
  if (d1 < 0.0) return NAN;
  return sqrt(d1);
}

/* ------------------------------------------------------------------ */
/* Uab is stored column-major here: Uab[idx*n + i] (the reference keeps it
 * n x n_index row-major; only the element values matter).               */

/* src/lmm.cpp:1213-1256  CalcUab(UtW,Uty,Uab): SNP-independent columns */
void orc_CalcUab_null(size_t n, size_t c, const double *UtW /* n x c row-major */,
                      const double *Uty, double *Uab) {
  for (size_t a = 1; a <= c + 2; ++a) {
    if (a == c + 1) continue;
    for (size_t b = a; b >= 1; --b) {
      if (b == c + 1) continue;
      size_t idx = orc_GetabIndex(a, b, c);
      double *col = Uab + idx * n;
      for (size_t i = 0; i < n; ++i) {
        double ua = (a == c + 2) ? Uty[i] : UtW[i * c + (a - 1)];
        double ub = (b == c + 2) ? Uty[i] : UtW[i * c + (b - 1)];
        col[i] = ub * ua;
      }
    }
  }
}

/* src/lmm.cpp:1258-1280  CalcUab(UtW,Uty,Utx,Uab): the c+2 columns (c+1,b) */
void orc_CalcUab_snp(size_t n, size_t c, const double *UtW, const double *Uty,
                     const double *Utx, double *Uab) {
  for (size_t b = 1; b <= c + 2; ++b) {
    size_t idx = orc_GetabIndex(c + 1, b, c);
    double *col = Uab + idx * n;
    for (size_t i = 0; i < n; ++i) {
      double ub = (b == c + 2) ? Uty[i] : (b == c + 1) ? Utx[i] : UtW[i * c + (b - 1)];
      col[i] = ub * Utx[i];
    }
  }
}

static double dotp(const double *w, const double *v, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += w[i] * v[i];
  return s;
}

/* src/lmm.cpp:283-357 CalcPab, :359-416 CalcPPab, :418-482 CalcPPPab.
 * Row 0 = weighted sums with H^k, rows p>=1 = projection recursion; the
 * update is skipped when ps_ww == 0 (:342,:399,:462). order = 1,2,3.     */
static void calc_pab_family(size_t n, size_t c, int order, const double *Hi,
                            const double *HiHi, const double *HiHiHi,
                            const double *Uab, double *Pab, double *PPab,
                            double *PPPab) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  for (size_t p = 0; p <= c + 1; ++p) {
    for (size_t a = p + 1; a <= c + 2; ++a) {
      for (size_t b = a; b <= c + 2; ++b) {
        size_t iab = orc_GetabIndex(a, b, c);
        if (p == 0) {
          const double *col = Uab + iab * n;
          Pab[iab] = dotp(Hi, col, n);
          if (order >= 2) PPab[iab] = dotp(HiHi, col, n);
          if (order >= 3) PPPab[iab] = dotp(HiHiHi, col, n);
        } else {
          size_t iaw = orc_GetabIndex(a, p, c);
          size_t ibw = orc_GetabIndex(b, p, c);
          size_t iww = orc_GetabIndex(p, p, c);
          const double *P0 = Pab + (p - 1) * n_index;
          double ps_ab = P0[iab], ps_aw = P0[iaw], ps_bw = P0[ibw], ps_ww = P0[iww];
          double r1;
          if (ps_ww != 0)
            r1 = ps_ab - ps_aw * ps_bw / ps_ww;
          else
            r1 = ps_ab;
          if (order >= 2) {
            const double *Q0 = PPab + (p - 1) * n_index;
            double ps2_ab = Q0[iab], ps2_aw = Q0[iaw], ps2_bw = Q0[ibw], ps2_ww = Q0[iww];
            double r2;
            if (ps_ww != 0) {
              r2 = ps2_ab + ps_aw * ps_bw * ps2_ww / (ps_ww * ps_ww);
              r2 -= (ps_aw * ps2_bw + ps_bw * ps2_aw) / ps_ww;
            } else {
              r2 = ps2_ab;
            }
            if (order >= 3) {
              const double *R0 = PPPab + (p - 1) * n_index;
              double ps3_ab = R0[iab], ps3_aw = R0[iaw], ps3_bw = R0[ibw], ps3_ww = R0[iww];
              double r3;
              if (ps_ww != 0) {
                r3 = ps3_ab - ps_aw * ps_bw * ps2_ww * ps2_ww / (ps_ww * ps_ww * ps_ww);
                r3 -= (ps_aw * ps3_bw + ps_bw * ps3_aw + ps2_aw * ps2_bw) / ps_ww;
                r3 += (ps_aw * ps2_bw * ps2_ww + ps_bw * ps2_aw * ps2_ww +
                       ps_aw * ps_bw * ps3_ww) /
                      (ps_ww * ps_ww);
              } else {
                r3 = ps3_ab;
              }
              PPPab[p * n_index + iab] = r3;
            }
            PPab[p * n_index + iab] = r2;
          }
          Pab[p * n_index + iab] = r1;
        }
      }
    }
  }
}

/* FUNC_PARAM, src/lmm.h:35-45 (e_mode is always 0 on this path; ab unused) */
typedef struct {
  int calc_null;
  size_t ni_test, n_cvt;
  const double *eval;
  const double *Uab; /* column-major n x n_index */
  /* scratch (3n doubles + 3*(c+2)*n_index) owned by the caller */
  double *Hi, *HiHi, *HiHiHi;
  double *Pab, *PPab, *PPPab;
  long n_eval; /* number of likelihood/derivative evaluations (diagnostic) */
} orc_func_param;

#define ORC_MAXC 16
#define ORC_MAXIDX ((ORC_MAXC + 3) * (ORC_MAXC + 2) / 2)

/* Hi = 1/(l*eval+1) (e_mode==0: src/lmm.cpp:504-512 and twins). Returns
 * sum log|l*eval+1| when want_logdet.                                    */
static double fill_H(orc_func_param *p, double l, int order, int want_logdet) {
  size_t n = p->ni_test;
  double logdet = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double v = p->eval[i] * l;
    v = v + 1.0;
    double h = 1.0 / v;
    p->Hi[i] = h;
    if (order >= 2) p->HiHi[i] = h * h;
    if (order >= 3) p->HiHiHi[i] = (h * h) * h;
    if (want_logdet) logdet += log(fabs(v));
  }
  return logdet;
}

static double sumv(const double *v, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += v[i];
  return s;
}

/* src/lmm.cpp:484-542 LogL_f */
double orc_LogL_f(double l, orc_func_param *p) {
  size_t c = p->n_cvt, n = p->ni_test, n_index = (c + 3) * (c + 2) / 2;
  size_t nc_total = p->calc_null ? c : c + 1;
  p->n_eval++;
  double logdet_h = fill_H(p, l, 1, 1);
  calc_pab_family(n, c, 1, p->Hi, 0, 0, p->Uab, p->Pab, 0, 0);
  double cst = 0.5 * (double)n * (log((double)n) - log(2 * M_PI) - 1.0);
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  double P_yy = p->Pab[nc_total * n_index + iyy];
  if (P_yy >= 0.0 && P_yy < ORC_P_YY_MIN) P_yy = ORC_P_YY_MIN; /* :527 */
  return cst - 0.5 * logdet_h - 0.5 * (double)n * log(P_yy);
}

/* src/lmm.cpp:544-640 LogL_dev1, :642-717 LogL_dev2, :719-797 LogL_dev12 */
void orc_LogL_dev12(double l, orc_func_param *p, int order, double *dev1, double *dev2) {
  size_t c = p->n_cvt, n = p->ni_test, n_index = (c + 3) * (c + 2) / 2;
  size_t nc_total = p->calc_null ? c : c + 1;
  p->n_eval++;
  fill_H(p, l, order, 0);
  double trace_Hi = sumv(p->Hi, n);
  double trace_HiHi = (order >= 3) ? sumv(p->HiHi, n) : 0.0;
  calc_pab_family(n, c, order, p->Hi, p->HiHi, p->HiHiHi, p->Uab, p->Pab, p->PPab, p->PPPab);
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  double P_yy = p->Pab[nc_total * n_index + iyy];
  double PP_yy = p->PPab[nc_total * n_index + iyy];
  double yPKPy = (P_yy - PP_yy) / l;
  double trace_HiK = ((double)n - trace_Hi) / l;
  if (dev1) *dev1 = -0.5 * trace_HiK + 0.5 * (double)n * yPKPy / P_yy;
  if (order >= 3 && dev2) {
    double PPP_yy = p->PPPab[nc_total * n_index + iyy];
    double trace_HiKHiK = ((double)n + trace_HiHi - 2 * trace_Hi) / (l * l);
    double yPKPKPy = (P_yy + PPP_yy - 2.0 * PP_yy) / (l * l);
    *dev2 = 0.5 * trace_HiKHiK -
            0.5 * (double)n * (2.0 * yPKPKPy * P_yy - yPKPy * yPKPy) / (P_yy * P_yy);
  }
}

/* src/lmm.cpp:799-864 LogRL_f */
double orc_LogRL_f(double l, orc_func_param *p) {
  size_t c = p->n_cvt, n = p->ni_test, n_index = (c + 3) * (c + 2) / 2;
  size_t nc_total;
  double df;
  if (p->calc_null) {
    nc_total = c;
    df = (double)n - (double)c;
  } else {
    nc_total = c + 1;
    df = (double)n - (double)c - 1.0;
  }
  p->n_eval++;
  double logdet_h = fill_H(p, l, 1, 1);
  calc_pab_family(n, c, 1, p->Hi, 0, 0, p->Uab, p->Pab, 0, 0);
  /* Iab = CalcPab with Hi == 1 (:839-840); PPab scratch reused for it */
  for (size_t i = 0; i < n; ++i) p->HiHi[i] = 1.0;
  calc_pab_family(n, c, 1, p->HiHi, 0, 0, p->Uab, p->PPab, 0, 0);
  double logdet_hiw = 0.0;
  for (size_t i = 0; i < nc_total; ++i) {
    size_t iww = orc_GetabIndex(i + 1, i + 1, c);
    logdet_hiw += log(p->Pab[i * n_index + iww]);
    logdet_hiw -= log(p->PPab[i * n_index + iww]);
  }
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  double P_yy = p->Pab[nc_total * n_index + iyy];
  if (P_yy >= 0.0 && P_yy < ORC_P_YY_MIN) P_yy = ORC_P_YY_MIN; /* :854 */
  double cst = 0.5 * df * (log(df) - log(2 * M_PI) - 1.0);
  return cst - 0.5 * logdet_h - 0.5 * logdet_hiw - 0.5 * df * log(P_yy);
}

/* src/lmm.cpp:866-943 LogRL_dev1, :945-1033 LogRL_dev2, :1035-1125 LogRL_dev12 */
void orc_LogRL_dev12(double l, orc_func_param *p, int order, double *dev1, double *dev2) {
  size_t c = p->n_cvt, n = p->ni_test, n_index = (c + 3) * (c + 2) / 2;
  size_t nc_total;
  double df;
  if (p->calc_null) {
    nc_total = c;
    df = (double)n - (double)c;
  } else {
    nc_total = c + 1;
    df = (double)n - (double)c - 1.0;
  }
  p->n_eval++;
  fill_H(p, l, order, 0);
  double trace_Hi = sumv(p->Hi, n);
  double trace_HiHi = (order >= 3) ? sumv(p->HiHi, n) : 0.0;
  calc_pab_family(n, c, order, p->Hi, p->HiHi, p->HiHiHi, p->Uab, p->Pab, p->PPab, p->PPPab);
  double trace_P = trace_Hi, trace_PP = trace_HiHi;
  for (size_t i = 0; i < nc_total; ++i) {
    size_t iww = orc_GetabIndex(i + 1, i + 1, c);
    double ps_ww = p->Pab[i * n_index + iww];
    double ps2_ww = p->PPab[i * n_index + iww];
    trace_P -= ps2_ww / ps_ww;
    if (order >= 3) {
      double ps3_ww = p->PPPab[i * n_index + iww];
      trace_PP += ps2_ww * ps2_ww / (ps_ww * ps_ww) - 2.0 * ps3_ww / ps_ww;
    }
  }
  double trace_PK = (df - trace_P) / l;
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  double P_yy = p->Pab[nc_total * n_index + iyy];
  double PP_yy = p->PPab[nc_total * n_index + iyy];
  double yPKPy = (P_yy - PP_yy) / l;
  if (dev1) *dev1 = -0.5 * trace_PK + 0.5 * df * yPKPy / P_yy;
  if (order >= 3 && dev2) {
    double PPP_yy = p->PPPab[nc_total * n_index + iyy];
    double trace_PKPK = (df + trace_PP - 2.0 * trace_P) / (l * l);
    double yPKPKPy = (P_yy + PPP_yy - 2.0 * PP_yy) / (l * l);
    *dev2 = 0.5 * trace_PKPK -
            0.5 * df * (2.0 * yPKPKPy * P_yy - yPKPy * yPKPy) / (P_yy * P_yy);
  }
}

static double f_dev1(char fn, double l, orc_func_param *p) {
  double d1;
  if (fn == 'R')
    orc_LogRL_dev12(l, p, 2, &d1, 0);
  else
    orc_LogL_dev12(l, p, 2, &d1, 0);
  return d1;
}
static void f_dev12(char fn, double l, orc_func_param *p, double *d1, double *d2) {
  if (fn == 'R')
    orc_LogRL_dev12(l, p, 3, d1, d2);
  else
    orc_LogL_dev12(l, p, 3, d1, d2);
}
static double f_logf(char fn, double l, orc_func_param *p) {
  return fn == 'R' ? orc_LogRL_f(l, p) : orc_LogL_f(l, p);
}

/* ---------------- GSL root finders, restated ----------------------- */
enum { ORC_SUCCESS = 0, ORC_CONTINUE = -2, ORC_EINVAL = 4, ORC_EBADFUNC = 9, ORC_EZERODIV = 12 };

typedef struct {
  double a, b, c, d, e, fa, fb, fc;
  double root, x_lower, x_upper;
} orc_brent;

/* GSL roots/brent.c brent_init (via gsl_root_fsolver_set, roots/fsolver.c) */
static int brent_set(orc_brent *s, char fn, orc_func_param *p, double x_lower, double x_upper) {
  if (x_lower > x_upper) return ORC_EINVAL;
  s->root = 0.5 * (x_lower + x_upper);
  s->x_lower = x_lower;
  s->x_upper = x_upper;
  double f_lower = f_dev1(fn, x_lower, p);
  if (!isfinite(f_lower)) return ORC_EBADFUNC;
  double f_upper = f_dev1(fn, x_upper, p);
  if (!isfinite(f_upper)) return ORC_EBADFUNC;
  s->a = x_lower;
  s->fa = f_lower;
  s->b = x_upper;
  s->fb = f_upper;
  s->c = x_upper;
  s->fc = f_upper;
  s->d = x_upper - x_lower;
  s->e = x_upper - x_lower;
  if ((f_lower < 0.0 && f_upper < 0.0) || (f_lower > 0.0 && f_upper > 0.0)) return ORC_EINVAL;
  return ORC_SUCCESS;
}

/* GSL roots/brent.c brent_iterate */
static int brent_iterate(orc_brent *s, char fn, orc_func_param *p) {
  double tol, m;
  int ac_equal = 0;
  double a = s->a, b = s->b, c = s->c, fa = s->fa, fb = s->fb, fc = s->fc, d = s->d, e = s->e;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) {
    ac_equal = 1;
    c = a;
    fc = fa;
    d = b - a;
    e = b - a;
  }
  if (fabs(fc) < fabs(fb)) {
    ac_equal = 1;
    a = b;
    b = c;
    c = a;
    fa = fb;
    fb = fc;
    fc = fa;
  }
  tol = 0.5 * DBL_EPSILON * fabs(b);
  m = 0.5 * (c - b);
  if (fb == 0) {
    s->root = b;
    s->x_lower = b;
    s->x_upper = b;
    return ORC_SUCCESS;
  }
  if (fabs(m) <= tol) {
    s->root = b;
    if (b < c) {
      s->x_lower = b;
      s->x_upper = c;
    } else {
      s->x_lower = c;
      s->x_upper = b;
    }
    return ORC_SUCCESS;
  }
  if (fabs(e) < tol || fabs(fa) <= fabs(fb)) {
    d = m; /* bisection */
    e = m;
  } else {
    double pp, q, r;
    double sv = fb / fa;
    if (ac_equal) {
      pp = 2 * m * sv;
      q = 1 - sv;
    } else {
      q = fa / fc;
      r = fb / fc;
      pp = sv * (2 * m * q * (q - r) - (b - a) * (r - 1));
      q = (q - 1) * (r - 1) * (sv - 1);
    }
    if (pp > 0)
      q = -q;
    else
      pp = -pp;
    double lim1 = 3 * m * q - fabs(tol * q), lim2 = fabs(e * q);
    if (2 * pp < (lim1 < lim2 ? lim1 : lim2)) {
      e = d;
      d = pp / q;
    } else {
      d = m;
      e = m;
    }
  }
  a = b;
  fa = fb;
  if (fabs(d) > tol)
    b += d;
  else
    b += (m > 0 ? +tol : -tol);
  fb = f_dev1(fn, b, p);
  if (!isfinite(fb)) return ORC_EBADFUNC; /* SAFE_FUNC_CALL: state not saved */
  s->a = a;
  s->b = b;
  s->c = c;
  s->d = d;
  s->e = e;
  s->fa = fa;
  s->fb = fb;
  s->fc = fc;
  s->root = b;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) c = a;
  if (b < c) {
    s->x_lower = b;
    s->x_upper = c;
  } else {
    s->x_lower = c;
    s->x_upper = b;
  }
  return ORC_SUCCESS;
}

/* GSL roots/convergence.c gsl_root_test_interval */
static int test_interval(double x_lower, double x_upper, double epsabs, double epsrel) {
  double abs_lower = fabs(x_lower), abs_upper = fabs(x_upper), min_abs, tolerance;
  if (x_lower > x_upper) return ORC_EINVAL;
  if ((x_lower > 0.0 && x_upper > 0.0) || (x_lower < 0.0 && x_upper < 0.0))
    min_abs = abs_lower < abs_upper ? abs_lower : abs_upper;
  else
    min_abs = 0;
  tolerance = epsabs + epsrel * min_abs;
  if (fabs(x_upper - x_lower) < tolerance) return ORC_SUCCESS;
  return ORC_CONTINUE;
}
/* GSL roots/convergence.c gsl_root_test_delta */
static int test_delta(double x1, double x0, double epsabs, double epsrel) {
  double tolerance = epsabs + epsrel * fabs(x1);
  if (fabs(x1 - x0) < tolerance || x1 == x0) return ORC_SUCCESS;
  return ORC_CONTINUE;
}

/* src/lmm.cpp:1945-2140 CalcLambda(func_name, params, ...).
 * diag (optional, 3 longs): [0] brent iterations, [1] newton iterations,
 * [2] number of sign-change brackets. */
void orc_CalcLambda(char func_name, orc_func_param *params, double l_min, double l_max,
                    size_t n_region, double *lambda, double *logf, long *diag) {
  *logf = NAN;
  *lambda = NAN;
  char fn = (func_name == 'R' || func_name == 'r') ? 'R' : 'L';
  if (func_name != 'R' && func_name != 'L' && func_name != 'r' && func_name != 'l') return;
  if (diag) diag[0] = diag[1] = diag[2] = 0;

  double *lo = (double *)malloc(sizeof(double) * 2 * (n_region + 1));
  double *hi = lo + n_region + 1;
  size_t nb = 0;
  double lambda_interval = log(l_max / l_min) / (double)n_region;
  double lambda_l, lambda_h, dev1_l, dev1_h, logf_l, logf_h;
  for (size_t i = 0; i < n_region; ++i) { /* :1967-1982 */
    lambda_l = l_min * exp(lambda_interval * i);
    lambda_h = l_min * exp(lambda_interval * (i + 1.0));
    dev1_l = f_dev1(fn, lambda_l, params);
    dev1_h = f_dev1(fn, lambda_h, params);
    if (dev1_l * dev1_h <= 0) {
      lo[nb] = lambda_l;
      hi[nb] = lambda_h;
      nb++;
    }
  }
  if (diag) diag[2] = (long)nb;

  if (nb == 0) { /* :1985-2000 */
    logf_l = f_logf(fn, l_min, params);
    logf_h = f_logf(fn, l_max, params);
    if (logf_l >= logf_h) {
      *lambda = l_min;
      *logf = logf_l;
    } else {
      *lambda = l_max;
      *logf = logf_h;
    }
    free(lo);
    return;
  }

  double l = 0.0, l_temp = 0.0;
  orc_brent bs;
  memset(&bs, 0, sizeof bs);
  for (size_t i = 0; i < nb; ++i) { /* :2029 */
    lambda_l = lo[i];
    lambda_h = hi[i];
    int set_status = brent_set(&bs, fn, params, lambda_l, lambda_h); /* return value ignored at :2034 */
    (void)set_status;
    int status;
    unsigned iter = 0;
    const unsigned max_iter = 100;
    do { /* :2040-2055 */
      iter++;
      status = brent_iterate(&bs, fn, params);
      if (diag) diag[0]++;
      if (status != ORC_SUCCESS && status != ORC_CONTINUE) break;
      l = bs.root;
      lambda_l = bs.x_lower;
      lambda_h = bs.x_upper;
      status = test_interval(lambda_l, lambda_h, 0, 1e-1);
      if (status != ORC_SUCCESS && status != ORC_CONTINUE) break;
    } while (status == ORC_CONTINUE && iter < max_iter);
    if (status == ORC_CONTINUE) break; /* :2057-2060 */

    /* Newton (GSL roots/newton.c): set evaluates fdf at the start */
    unsigned iter2 = 0;
    double nf, ndf, root = l;
    f_dev12(fn, root, params, &nf, &ndf);
    do { /* :2064-2078 */
      iter2++;
      if (diag) diag[1]++;
      if (ndf == 0.0) {
        status = ORC_EZERODIV;
      } else {
        double root_new = root - (nf / ndf);
        root = root_new;
        double f_new, df_new;
        f_dev12(fn, root_new, params, &f_new, &df_new);
        nf = f_new;
        ndf = df_new;
        if (!isfinite(f_new) || !isfinite(df_new))
          status = ORC_EBADFUNC;
        else
          status = ORC_SUCCESS;
      }
      if (status != ORC_SUCCESS && status != ORC_CONTINUE) break;
      l_temp = l;
      l = root;
      status = test_delta(l, l_temp, 0, 1e-5);
      if (status != ORC_SUCCESS && status != ORC_CONTINUE) break;
    } while (status == ORC_CONTINUE && iter2 < max_iter && l > l_min && l < l_max);

    if (status == ORC_CONTINUE || status != ORC_SUCCESS) { /* :2087-2094 */
      *logf = NAN;
      *lambda = NAN;
      free(lo);
      return;
    }
    l = l_temp; /* :2096 the previous Newton iterate is what gets reported */
    if (l < l_min) l = l_min;
    if (l > l_max) l = l_max;
    logf_l = f_logf(fn, l, params);
    if (i == 0) {
      *logf = logf_l;
      *lambda = l;
    } else if (*logf < logf_l) {
      *logf = logf_l;
      *lambda = l;
    }
  }
  logf_l = f_logf(fn, l_min, params); /* :2121-2136 */
  logf_h = f_logf(fn, l_max, params);
  if (logf_l > *logf) {
    *lambda = l_min;
    *logf = logf_l;
  }
  if (logf_h > *logf) {
    *lambda = l_max;
    *logf = logf_h;
  }
  free(lo);
}

/* ---------------- GSL cdf pieces, restated ------------------------- */
/* GSL cdf/beta_inc.c beta_cont_frac: modified Lentz on the continued
 * fraction for I_x(a,b), <= 512 double-steps. */
static double beta_cont_frac(double a, double b, double x, double epsabs) {
  const unsigned max_iter = 512;
  const double cutoff = 2.0 * DBL_MIN;
  unsigned it = 0;
  double num = 1.0;
  double den = 1.0 - (a + b) * x / (a + 1.0);
  if (fabs(den) < cutoff) den = NAN;
  den = 1.0 / den;
  double cf = den;
  while (it < max_iter) {
    const int k = (int)it + 1;
    double coeff = k * (b - k) * x / (((a - 1.0) + 2 * k) * (a + 2 * k));
    double delta;
    den = 1.0 + coeff * den;
    num = 1.0 + coeff / num;
    if (fabs(den) < cutoff) den = NAN;
    if (fabs(num) < cutoff) num = NAN;
    den = 1.0 / den;
    delta = den * num;
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
    ++it;
  }
  if (it >= max_iter) return NAN;
  return cf;
}

/* upper/lower regularised incomplete gamma for the only shape needed here,
 * a = 1/2:  Q(1/2,z) = erfc(sqrt z),  P(1/2,z) = erf(sqrt z). */
static double gamma_half_Q(double z) { return erfc(sqrt(z)); }
static double gamma_half_P(double z) { return erf(sqrt(z)); }

/* GSL cdf/beta_inc.c beta_inc_AXPY: A*I_x(a,b)+Y. The two asymptotic branches
 * call gsl_sf_gamma_inc_Q/P(b or a, .); on this path the small parameter is
 * always nu1/2 = 1/2, anything else falls through to the continued fraction. */
static double beta_inc_AXPY(double A, double Y, double a, double b, double x) {
  if (x == 0.0) return A * 0 + Y;
  if (x == 1.0) return A * 1 + Y;
  if (a > 1e5 && b < 10 && x > a / (a + b) && b == 0.5) {
    double N = a + (b - 1.0) / 2.0;
    return A * gamma_half_Q(-N * log(x)) + Y;
  }
  if (b > 1e5 && a < 10 && x < b / (a + b) && a == 0.5) {
    double N = b + (a - 1.0) / 2.0;
    return A * gamma_half_P(-N * log1p(-x)) + Y;
  }
  double ln_beta = lgamma(a) + lgamma(b) - lgamma(a + b);
  double ln_pre = -ln_beta + a * log(x) + b * log1p(-x);
  double prefactor = exp(ln_pre);
  if (x < (a + 1.0) / (a + b + 2.0)) {
    double epsabs = fabs(Y / (A * prefactor / a)) * DBL_EPSILON;
    double cf = beta_cont_frac(a, b, x, epsabs);
    return A * (prefactor * cf / a) + Y;
  } else {
    double epsabs = fabs((A + Y) / (A * prefactor / b)) * DBL_EPSILON;
    double cf = beta_cont_frac(b, a, 1.0 - x, epsabs);
    double term = prefactor * cf / b;
    if (A == -Y) return -A * term;
    return A * (1 - term) + Y;
  }
}

/* GSL cdf/fdist.c gsl_cdf_fdist_Q (called at src/lmm.cpp:1161,1206) */
double orc_cdf_fdist_Q(double x, double nu1, double nu2) {
  double r = nu2 / nu1;
  if (x < r) {
    double u = x / (r + x);
    return beta_inc_AXPY(-1.0, 1.0, nu1 / 2.0, nu2 / 2.0, u);
  } else {
    double u = r / (r + x);
    return beta_inc_AXPY(1.0, 0.0, nu2 / 2.0, nu1 / 2.0, u);
  }
}

/* GSL cdf/chisq.c -> cdf/gamma.c gsl_cdf_gamma_Q(x, nu/2, 2) for nu == 1
 * (called at src/lmm.cpp:1553) */
double orc_cdf_chisq_Q1(double x) {
  if (x <= 0.0) return 1.0;
  double y = x / 2.0;
  if (y < 0.5) return 1.0 - gamma_half_P(y);
  return gamma_half_Q(y);
}

/* ---------------- Wald / score ------------------------------------- */
/* src/lmm.cpp:1127-1167 CalcRLWald; :1170-1211 CalcRLScore */
static void wald_or_score(int score, double l, orc_func_param *p, size_t ni_test,
                          double *beta, double *se, double *pval) {
  size_t c = p->n_cvt, n = p->ni_test, n_index = (c + 3) * (c + 2) / 2;
  int df = (int)ni_test - (int)c - 1;
  p->n_eval++;
  fill_H(p, l, 1, 0);
  calc_pab_family(n, c, 1, p->Hi, 0, 0, p->Uab, p->Pab, 0, 0);
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  size_t ixx = orc_GetabIndex(c + 1, c + 1, c);
  size_t ixy = orc_GetabIndex(c + 2, c + 1, c);
  double P_yy = p->Pab[c * n_index + iyy];
  double P_xx = p->Pab[c * n_index + ixx];
  double P_xy = p->Pab[c * n_index + ixy];
  double Px_yy = p->Pab[(c + 1) * n_index + iyy];
  *beta = P_xy / P_xx;
  double tau = (double)df / Px_yy;
  *se = orc_safe_sqrt(1.0 / (tau * P_xx));
  if (score)
    *pval = orc_cdf_fdist_Q((double)ni_test * P_xy * P_xy / (P_yy * P_xx), 1.0, df);
  else
    *pval = orc_cdf_fdist_Q((P_yy - Px_yy) * tau, 1.0, df);
}

static int param_alloc(orc_func_param *p, size_t n, size_t c) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  p->Hi = (double *)malloc(sizeof(double) * (3 * n + 3 * (c + 2) * n_index));
  if (!p->Hi) return -1;
  p->HiHi = p->Hi + n;
  p->HiHiHi = p->HiHi + n;
  p->Pab = p->HiHiHi + n;
  p->PPab = p->Pab + (c + 2) * n_index;
  p->PPPab = p->PPab + (c + 2) * n_index;
  p->n_eval = 0;
  return 0;
}

/* src/lmm.cpp:2143-2180 CalcLambda (null model, calc_null = true) */
void orc_CalcLambda_null(char func_name, size_t n, size_t c, const double *eval,
                         const double *UtW, const double *Uty, double l_min, double l_max,
                         size_t n_region, double *lambda, double *logl_H0) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_CalcUab_null(n, c, UtW, Uty, Uab);
  orc_func_param p = {1, n, c, eval, Uab};
  param_alloc(&p, n, c);
  orc_CalcLambda(func_name, &p, l_min, l_max, n_region, lambda, logl_H0, 0);
  free(p.Hi);
  free(Uab);
}

/* src/lmm.cpp:2183-2205 CalcPve */
void orc_CalcPve(size_t n, size_t c, const double *eval, const double *UtW, const double *Uty,
                 double lambda, double trace_G, double *pve, double *pve_se) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_CalcUab_null(n, c, UtW, Uty, Uab);
  orc_func_param p = {1, n, c, eval, Uab};
  param_alloc(&p, n, c);
  double d1, d2;
  orc_LogRL_dev12(lambda, &p, 3, &d1, &d2);
  double se = orc_safe_sqrt(-1.0 / d2);
  *pve = trace_G * lambda / (trace_G * lambda + 1.0);
  *pve_se = trace_G / ((trace_G * lambda + 1.0) * (trace_G * lambda + 1.0)) * se;
  free(p.Hi);
  free(Uab);
}

/* small dense LU with partial pivoting (GSL linalg/lu.c, used through
 * src/lapack.cpp:307-352) -- c x c systems only */
static int lu_decomp(double *A, size_t m, size_t *perm) {
  for (size_t i = 0; i < m; ++i) perm[i] = i;
  for (size_t j = 0; j + 1 < m; ++j) {
    double amax = fabs(A[j * m + j]);
    size_t ip = j;
    for (size_t i = j + 1; i < m; ++i)
      if (fabs(A[i * m + j]) > amax) {
        amax = fabs(A[i * m + j]);
        ip = i;
      }
    if (ip != j) {
      for (size_t k = 0; k < m; ++k) {
        double t = A[j * m + k];
        A[j * m + k] = A[ip * m + k];
        A[ip * m + k] = t;
      }
      size_t t = perm[j];
      perm[j] = perm[ip];
      perm[ip] = t;
    }
    double ajj = A[j * m + j];
    if (ajj != 0.0)
      for (size_t i = j + 1; i < m; ++i) {
        double aij = A[i * m + j] / ajj;
        A[i * m + j] = aij;
        for (size_t k = j + 1; k < m; ++k) A[i * m + k] -= aij * A[j * m + k];
      }
  }
  return 0;
}
static void lu_solve(const double *LU, size_t m, const size_t *perm, const double *b, double *x) {
  for (size_t i = 0; i < m; ++i) x[i] = b[perm[i]];
  for (size_t i = 0; i < m; ++i)
    for (size_t k = 0; k < i; ++k) x[i] -= LU[i * m + k] * x[k];
  for (size_t ii = m; ii-- > 0;) {
    for (size_t k = ii + 1; k < m; ++k) x[ii] -= LU[ii * m + k] * x[k];
    x[ii] /= LU[ii * m + ii];
  }
}

/* src/lmm.cpp:2210-2281 CalcLmmVgVeBeta */
void orc_CalcLmmVgVeBeta(size_t n, size_t c, const double *eval, const double *UtW,
                         const double *Uty, double lambda, double *vg, double *ve,
                         double *beta, double *se_beta) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_CalcUab_null(n, c, UtW, Uty, Uab);
  orc_func_param p = {1, n, c, eval, Uab};
  param_alloc(&p, n, c);
  fill_H(&p, lambda, 1, 0);
  double *WHiW = (double *)calloc(c * c + 2 * c + c * c, sizeof(double));
  double *WHiy = WHiW + c * c, *rhs = WHiy + c, *Vbeta = rhs + c;
  for (size_t a = 0; a < c; ++a) {
    for (size_t b = 0; b < c; ++b) {
      double s = 0;
      for (size_t i = 0; i < n; ++i) s += (UtW[i * c + a] * p.Hi[i]) * UtW[i * c + b];
      WHiW[a * c + b] = s;
    }
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += (UtW[i * c + a] * p.Hi[i]) * Uty[i];
    WHiy[a] = s;
  }
  size_t perm[ORC_MAXC];
  lu_decomp(WHiW, c, perm);
  lu_solve(WHiW, c, perm, WHiy, beta);
  for (size_t j = 0; j < c; ++j) { /* LUInvert: solve for unit vectors */
    for (size_t k = 0; k < c; ++k) rhs[k] = (k == j) ? 1.0 : 0.0;
    double col[ORC_MAXC];
    lu_solve(WHiW, c, perm, rhs, col);
    for (size_t k = 0; k < c; ++k) Vbeta[k * c + j] = col[k];
  }
  calc_pab_family(n, c, 1, p.Hi, 0, 0, Uab, p.Pab, 0, 0);
  size_t iyy = orc_GetabIndex(c + 2, c + 2, c);
  double P_yy = p.Pab[c * n_index + iyy];
  *ve = P_yy / (double)(n - c);
  *vg = *ve * lambda;
  for (size_t i = 0; i < c; ++i) se_beta[i] = orc_safe_sqrt(Vbeta[i * c + i] * (*ve));
  free(WHiW);
  free(p.Hi);
  free(Uab);
}

/* ---------------- per-batch association ---------------------------- */
/* The body of batch_compute, src/lmm.cpp:1526-1562 (BIMBAM) and
 * :1853-1888 (PLINK; plink_nan_rule != 0).  UtX is SNP-major here:
 * UtX[s*n + i] (the reference keeps SNPs as strided columns, :1528).
 * carry[2] = {beta, se} carried between SNPs, reproducing the PLINK loop's
 * function-scope variables (:1725) when CalcRLWald is skipped (:1870).   */
void orc_lmm_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtW,
                   const double *Uty, const double *UtX, size_t l, double l_min, double l_max,
                   size_t n_region, double l_mle_null, double logl_mle_H0, int plink_nan_rule,
                   double *carry, orc_sumstat *out, long *diag /* l*3 or NULL */) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_CalcUab_null(n, c, UtW, Uty, Uab); /* :1508 */
  orc_func_param p = {0, n, c, eval, Uab};
  param_alloc(&p, n, c);
  double cb = carry ? carry[0] : 0.0, cs = carry ? carry[1] : 0.0;
  for (size_t s = 0; s < l; ++s) {
    orc_CalcUab_snp(n, c, UtW, Uty, UtX + s * n, Uab); /* :1531 */
    double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
    double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0;
    if (plink_nan_rule) {
      beta = cb;
      se = cs;
    }
    if (a_mode == 3 || a_mode == 4 || a_mode == 9) /* :1541 */
      wald_or_score(1, l_mle_null, &p, n, &beta, &se, &p_score);
    if (a_mode == 1 || a_mode == 4) { /* :1545 */
      orc_CalcLambda('R', &p, l_min, l_max, n_region, &lambda_remle, &logl_H1, diag ? diag + 3 * s : 0);
      if (!plink_nan_rule || !isnan(logl_H1)) /* :1870 */
        wald_or_score(0, lambda_remle, &p, n, &beta, &se, &p_wald);
    }
    if (a_mode == 2 || a_mode == 4 || a_mode == 9) { /* :1551 */
      orc_CalcLambda('L', &p, l_min, l_max, n_region, &lambda_mle, &logl_H1,
                     (diag && a_mode != 4) ? diag + 3 * s : 0);
      p_lrt = orc_cdf_chisq_Q1(2.0 * (logl_H1 - logl_mle_H0));
      if (isnan(logl_H1)) p_lrt = NAN; /* gsl_cdf_gamma_Q(NaN): NaN propagates */
    }
    if (plink_nan_rule && isnan(logl_H1)) p_wald = p_lrt = logl_H1; /* :1882-1884 */
    out[s].beta = beta;
    out[s].se = se;
    out[s].lambda_remle = lambda_remle;
    out[s].lambda_mle = lambda_mle;
    out[s].p_wald = p_wald;
    out[s].p_lrt = p_lrt;
    out[s].p_score = p_score;
    out[s].logl_H1 = logl_H1;
    cb = beta;
    cs = se;
    if (plink_nan_rule) { /* PLINK loop: lambda/p variables are function scope too */
    }
  }
  if (carry) {
    carry[0] = cb;
    carry[1] = cs;
  }
  free(p.Hi);
  free(Uab);
}

/* Test aid for the two-tier lambda criterion: for every SNP row of UtX and a GIVEN lambda, the relative size
 * of the Newton step CalcLambda's polish would take from there, |f / f'| / |lambda - f / f'| with f = dev1, f' = dev2 of
 * LogRL ('R') or LogL ('L') (src/lmm.cpp:2064-2078), and logf(lambda).  The reference reports the iterate BEFORE the one that
 * met |x_new - x_old| < 1e-5 |x_new| (:2073, :2096): a lambda-hat that differs from the reference's because a Brent / Newton
 * trip count flipped is still a point where this step is below that threshold; a wrong root is not. */
void orc_newton_step_rel(char func_name, size_t n, size_t c, const double *eval, const double *UtW, const double *Uty,
                         const double *UtX, size_t l, const double *lambda, double *step_rel, double *logf) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_CalcUab_null(n, c, UtW, Uty, Uab);
  orc_func_param p = {0, n, c, eval, Uab};
  param_alloc(&p, n, c);
  char fn = (func_name == 'R' || func_name == 'r') ? 'R' : 'L';
  for (size_t s = 0; s < l; ++s) {
    orc_CalcUab_snp(n, c, UtW, Uty, UtX + s * n, Uab);
    double f, df;
    f_dev12(fn, lambda[s], &p, &f, &df);
    double x1 = lambda[s] - f / df;
    step_rel[s] = fabs(f / df) / fabs(x1);
    if (logf) logf[s] = f_logf(fn, lambda[s], &p);
  }
  free(p.Hi);
  free(Uab);
}

/* ---------------- AnalyzeGene --------------------------------------- */
/* LMM::AnalyzeGene, src/lmm.cpp:1365-1471: every row of UtY (l x n, SNP-major style) is a rotated PHENOTYPE U^T y_g,
 * Utx is the one fixed tested variable.  Per row: Uab from (UtW, Uty_g) with the x columns zero; param0 carries
 * calc_null = FALSE (:1422) -- the x step of the Pab recursion is then skipped because ps_ww == 0 (:342) -- and
 * CalcLambda('L') gives l_H0, logl_H0 (:1424-1427); then the x columns are filled (:1429) and the usual score (at
 * l_H0) / REML+Wald / ML+LRT (against logl_H0) follow (:1432-1450).  Restated with the same FUNC_PARAM quirk.      */
void orc_gene_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtW, const double *Utx,
                    const double *UtY, size_t l, double l_min, double l_max, size_t n_region, orc_sumstat *out) {
  size_t n_index = (c + 3) * (c + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  orc_func_param p = {0, n, c, eval, Uab};
  param_alloc(&p, n, c);
  for (size_t g = 0; g < l; ++g) {
    const double *Uty = UtY + g * n;
    double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
    double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0, logl_H0 = 0.0, l_H0 = 0.0;
    memset(Uab, 0, n * n_index * sizeof(double)); /* gsl_matrix_set_zero(Uab), :1419 */
    orc_CalcUab_null(n, c, UtW, Uty, Uab);         /* :1421 */
    if (a_mode == 2 || a_mode == 3 || a_mode == 4 || a_mode == 9) /* :1424 */
      orc_CalcLambda('L', &p, l_min, l_max, n_region, &l_H0, &logl_H0, 0);
    orc_CalcUab_snp(n, c, UtW, Uty, Utx, Uab); /* :1429 */
    if (a_mode == 3 || a_mode == 4 || a_mode == 9) wald_or_score(1, l_H0, &p, n, &beta, &se, &p_score);
    if (a_mode == 1 || a_mode == 4) {
      orc_CalcLambda('R', &p, l_min, l_max, n_region, &lambda_remle, &logl_H1, 0);
      wald_or_score(0, lambda_remle, &p, n, &beta, &se, &p_wald);
    }
    if (a_mode == 2 || a_mode == 4 || a_mode == 9) {
      orc_CalcLambda('L', &p, l_min, l_max, n_region, &lambda_mle, &logl_H1, 0);
      p_lrt = orc_cdf_chisq_Q1(2.0 * (logl_H1 - logl_H0));
      if (isnan(logl_H1) || isnan(logl_H0)) p_lrt = NAN;
    }
    out[g].beta = beta;
    out[g].se = se;
    out[g].lambda_remle = lambda_remle;
    out[g].lambda_mle = lambda_mle;
    out[g].p_wald = p_wald;
    out[g].p_lrt = p_lrt;
    out[g].p_score = p_score;
    out[g].logl_H1 = logl_H1;
  }
  free(p.Hi);
  free(Uab);
}

/* ---------------- GXE ----------------------------------------------- */
/* LMM::AnalyzeBimbamGXE :2283-2425 / AnalyzePlinkGXE :2427-2608, the per-SNP part (:2362-2408 / :2540-2589) on rotated
 * inputs: UtWe = [U^T W | U^T env] (n x (c+1) row-major), UtX rows = U^T x_s (x_s mean-imputed and recoded 2 - x when
 * x_mean > 1), UtZ rows = U^T (x_s . env), flip[s] = recoded.  Per SNP the covariate matrix UtW_expand = [UtWe | UtX_s]
 * has c + 2 columns; FUNC_PARAM.n_cvt = c + 2.  logl_H0 comes from CalcLambda('L') with calc_null = TRUE and is computed
 * for a_mode 2 and 4 ONLY (:2384) -- a_mode 9 compares against logl_H0 = 0, as the reference does.                  */
void orc_gxe_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtWe, const double *Uty,
                   const double *UtX, const double *UtZ, const int *flip, size_t l, double l_min, double l_max,
                   size_t n_region, double l_mle_null, orc_sumstat *out) {
  const size_t ce = c + 2;
  size_t n_index = (ce + 3) * (ce + 2) / 2;
  double *Uab = (double *)calloc(n * n_index, sizeof(double));
  double *UtW_expand = (double *)malloc(n * ce * sizeof(double));
  for (size_t i = 0; i < n; ++i)
    for (size_t a = 0; a < c + 1; ++a) UtW_expand[i * ce + a] = UtWe[i * (c + 1) + a];
  orc_func_param p = {0, n, ce, eval, Uab};
  param_alloc(&p, n, ce);
  for (size_t s = 0; s < l; ++s) {
    const double *Utx = UtX + s * n, *Utz = UtZ + s * n;
    for (size_t i = 0; i < n; ++i) UtW_expand[i * ce + (c + 1)] = Utx[i]; /* :2364 */
    double lambda_mle = 0.0, lambda_remle = 0.0, beta = 0.0, se = 0.0, p_wald = 0.0;
    double p_lrt = 0.0, p_score = 0.0, logl_H1 = 0.0, logl_H0 = 0.0;
    memset(Uab, 0, n * n_index * sizeof(double)); /* :2369 */
    orc_CalcUab_null(n, ce, UtW_expand, Uty, Uab); /* :2370 */
    if (a_mode == 2 || a_mode == 4) {              /* :2372-2375 */
      p.calc_null = 1;
      orc_CalcLambda('L', &p, l_min, l_max, n_region, &lambda_mle, &logl_H0, 0);
    }
    orc_CalcUab_snp(n, ce, UtW_expand, Uty, Utz, Uab); /* :2377 */
    p.calc_null = 0;
    if (a_mode == 3 || a_mode == 4 || a_mode == 9) wald_or_score(1, l_mle_null, &p, n, &beta, &se, &p_score);
    if (a_mode == 1 || a_mode == 4) {
      orc_CalcLambda('R', &p, l_min, l_max, n_region, &lambda_remle, &logl_H1, 0);
      wald_or_score(0, lambda_remle, &p, n, &beta, &se, &p_wald);
    }
    if (a_mode == 2 || a_mode == 4 || a_mode == 9) {
      orc_CalcLambda('L', &p, l_min, l_max, n_region, &lambda_mle, &logl_H1, 0);
      p_lrt = orc_cdf_chisq_Q1(2.0 * (logl_H1 - logl_H0));
      if (isnan(logl_H1) || isnan(logl_H0)) p_lrt = NAN;
    }
    if (flip[s]) beta *= -1; /* :2403 */
    out[s].beta = beta;
    out[s].se = se;
    out[s].lambda_remle = lambda_remle;
    out[s].lambda_mle = lambda_mle;
    out[s].p_wald = p_wald;
    out[s].p_lrt = p_lrt;
    out[s].p_score = p_score;
    out[s].logl_H1 = logl_H1;
  }
  free(p.Hi);
  free(UtW_expand);
  free(Uab);
}

/* ---------------- linear model (-lm) -------------------------------- */
/* LmCalcP, src/lm.cpp:266-287 */
static void lm_calc_p(int test_mode, double yPwy, double xPwy, double xPwx, double df, size_t n_size,
                      double *beta, double *se, double *p_wald, double *p_lrt, double *p_score) {
  double yPxy = yPwy - xPwy * xPwy / xPwx;
  double se_wald, se_score;
  *beta = xPwy / xPwx;
  se_wald = sqrt(yPxy / (df * xPwx));
  se_score = sqrt(yPwy / ((double)n_size * xPwx));
  *p_wald = orc_cdf_fdist_Q(*beta * *beta / (se_wald * se_wald), 1.0, df);
  *p_score = orc_cdf_fdist_Q(*beta * *beta / (se_score * se_score), 1.0, df);
  {
    double xl = (double)n_size * (log(yPwy) - log(yPxy));
    *p_lrt = isnan(xl) ? NAN : orc_cdf_chisq_Q1(xl);
  }
  *se = (test_mode == 3) ? se_score : se_wald;
}

/* The per-SNP body of LM::AnalyzeBimbam / AnalyzePlink, src/lm.cpp:382-640, with CalcvPv (:224-263):
 * X is SNP-major l x n, already mean-imputed; W n x c row-major; WtWi c x c; a_mode 51..54. */
void orc_lm_batch(int a_mode, size_t n, size_t c, const double *W, const double *WtWi, const double *y,
                  const double *X, size_t l, orc_sumstat *out) {
  double Wty[ORC_MAXC], Wtx[ORC_MAXC], t[ORC_MAXC];
  double yPwy = 0.0, d;
  double df = (double)n - (double)c - 1.0;
  for (size_t a = 0; a < c; ++a) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += W[i * c + a] * y[i];
    Wty[a] = s;
  }
  for (size_t i = 0; i < n; ++i) yPwy += y[i] * y[i];
  d = 0.0;
  for (size_t a = 0; a < c; ++a) {
    double s = 0.0;
    for (size_t b = 0; b < c; ++b) s += WtWi[a * c + b] * Wty[b];
    d += s * Wty[a];
  }
  yPwy -= d;
  for (size_t s_ = 0; s_ < l; ++s_) {
    const double *x = X + s_ * n;
    double xPwx = 0.0, xPwy = 0.0;
    for (size_t a = 0; a < c; ++a) {
      double s = 0.0;
      for (size_t i = 0; i < n; ++i) s += W[i * c + a] * x[i];
      Wtx[a] = s;
    }
    for (size_t i = 0; i < n; ++i) {
      xPwx += x[i] * x[i];
      xPwy += x[i] * y[i];
    }
    for (size_t a = 0; a < c; ++a) {
      double s = 0.0;
      for (size_t b = 0; b < c; ++b) s += WtWi[a * c + b] * Wtx[b];
      t[a] = s;
    }
    d = 0.0;
    for (size_t a = 0; a < c; ++a) d += t[a] * Wtx[a];
    xPwx -= d;
    d = 0.0;
    for (size_t a = 0; a < c; ++a) d += t[a] * Wty[a];
    xPwy -= d;
    double beta, se, p_wald, p_lrt, p_score;
    lm_calc_p(a_mode - 50, yPwy, xPwy, xPwx, df, n, &beta, &se, &p_wald, &p_lrt, &p_score);
    out[s_].beta = beta;
    out[s_].se = se;
    out[s_].lambda_remle = 0.0;
    out[s_].lambda_mle = 0.0;
    out[s_].p_wald = p_wald;
    out[s_].p_lrt = p_lrt;
    out[s_].p_score = p_score;
    out[s_].logl_H1 = -0.0;
  }
}

/* ---------------- genotype preparation ----------------------------- */
/* src/lmm.cpp:1590-1618 (BIMBAM) / :1779-1827 (PLINK): mean-impute only.
 * X is SNP-major l x n with NaN for missing, modified in place.          */
void orc_impute_mean(double *X, size_t l, size_t n) {
  for (size_t s = 0; s < l; ++s) {
    double *x = X + s * n;
    double tot = 0.0;
    size_t n_miss = 0;
    for (size_t i = 0; i < n; ++i) {
      if (isnan(x[i]))
        n_miss++;
      else
        tot += x[i];
    }
    double mean = tot / (double)(n - n_miss);
    for (size_t i = 0; i < n; ++i)
      if (isnan(x[i])) x[i] = mean;
  }
}

/* src/gemma_io.cpp:1487-1538 (BimbamKin) / :1651-1704 (PlinkKin): per SNP
 * over ALL ni_total individuals: mean over non-missing, impute, centre,
 * optional 1/sqrt(var) (k_mode 2, var formula :1511-1514). In place.     */
void orc_kin_prepare(double *X, size_t l, size_t n, int k_mode) {
  for (size_t s = 0; s < l; ++s) {
    double *x = X + s * n;
    double mean = 0.0, var = 0.0;
    size_t n_miss = 0;
    for (size_t i = 0; i < n; ++i) {
      if (isnan(x[i]))
        n_miss++;
      else {
        mean += x[i];
        var += x[i] * x[i];
      }
    }
    mean /= (double)(n - n_miss);
    var += mean * mean * (double)n_miss;
    var /= (double)n;
    var -= mean * mean;
    for (size_t i = 0; i < n; ++i)
      if (isnan(x[i])) x[i] = mean;
    for (size_t i = 0; i < n; ++i) x[i] += -1.0 * mean;
    if (k_mode == 2 && var != 0) {
      double sc = 1.0 / sqrt(var);
      for (size_t i = 0; i < n; ++i) x[i] *= sc;
    }
  }
}

/* PLINK .bed decode for one SNP (src/lmm.cpp:1783-1817,
 * src/gemma_io.cpp:1655-1686): 4 individuals per byte, low bits first;
 * (b0,b1): (0,0)->2, (0,1)->1, (1,1)->0, (1,0)->missing (NaN here).
 * indicator (ni_total ints, may be NULL = keep all) drops individuals.   */
size_t orc_bed_decode(const unsigned char *bytes, size_t ni_total, const int *indicator,
                      double *x) {
  size_t pos = 0;
  for (size_t i = 0; i < ni_total; ++i) {
    if (indicator && indicator[i] == 0) continue;
    unsigned b = bytes[i >> 2] >> (2 * (i & 3));
    unsigned b0 = b & 1u, b1 = (b >> 1) & 1u;
    double g;
    if (b0 == 0)
      g = (b1 == 0) ? 2.0 : 1.0;
    else
      g = (b1 == 1) ? 0.0 : NAN;
    x[pos++] = g;
  }
  return pos;
}

/* src/mathfunc.cpp:147-177 CenterMatrix: dgemv, dsyr2 and dsyr on the upper
 * triangle, then mirror to the lower one. */
void orc_CenterMatrix(double *G, size_t n) {
  double *Gw = (double *)malloc(sizeof(double) * n);
  for (size_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (size_t j = 0; j < n; ++j) s += G[i * n + j];
    Gw[i] = s;
  }
  double alpha = -1.0 / (double)n;
  for (size_t i = 0; i < n; ++i)
    for (size_t j = i; j < n; ++j) G[i * n + j] += alpha * Gw[i] + alpha * Gw[j];
  double d = 0.0;
  for (size_t i = 0; i < n; ++i) d += Gw[i];
  double beta = d / ((double)n * (double)n);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = i; j < n; ++j) G[i * n + j] += beta;
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < i; ++j) G[i * n + j] = G[j * n + i];
  free(Gw);
}

/* src/lapack.cpp:266-277 the eigenvalue post-processing of EigenDecomp_Zeroed */
double orc_zero_small_eval(double *eval, size_t n) {
  double d = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (eval[i] < 1e-10) eval[i] = 0.0;
    d += eval[i];
  }
  return d / (double)n;
}

/* Reference GEMM (row-major C = alpha*op(A)*op(B) + beta*C), the contract of
 * fast_cblas_dgemm src/fastblas.cpp:66-170; plain loops, for KATs only.   */
void orc_dgemm(char ta, char tb, size_t M, size_t N, size_t K, double alpha, const double *A,
               size_t lda, const double *B, size_t ldb, double beta, double *C, size_t ldc) {
  int tA = (ta == 'T' || ta == 't'), tB = (tb == 'T' || tb == 't');
  for (size_t i = 0; i < M; ++i)
    for (size_t j = 0; j < N; ++j) {
      double s = 0.0;
      for (size_t k = 0; k < K; ++k) {
        double a = tA ? A[k * lda + i] : A[i * lda + k];
        double b = tB ? B[j * ldb + k] : B[k * ldb + j];
        s += a * b;
      }
      C[i * ldc + j] = alpha * s + (beta == 0.0 ? 0.0 : beta * C[i * ldc + j]);
    }
}
