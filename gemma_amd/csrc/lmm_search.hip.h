// The lambda search of one SNP as plain C++ that compiles for the device (hipcc) AND for the host (g++: the CPU
// test harness tests/cpp/cheb_search_check.cpp runs this very code against the reference restatement under tests).
//
//  (1) GSL's root finders as CalcLambda uses them (GEMMA src/lmm.cpp:2024-2102): roots/brent.c, roots/newton.c and
//      roots/convergence.c restated over an evaluator object E { bool dev1(l, &d1); bool dev12(l, &d1, &d2); } whose
//      functions return false when they cannot evaluate at l (a table queried outside its interval) -- the bracket is
//      then abandoned with PB_OUTSIDE and the caller repeats it with the streaming evaluator, from scratch.
//  (2) Chebyshev-in-log(lambda) tables of the row-0 sums S_ab(t) = sum_i a_i b_i / (e^t delta_i + 1): every
//      H_i = 1 / (lambda delta_i + 1) is a logistic curve in t = log(lambda) whose nearest singularity lies pi off the real
//      axis, so on one grid interval of CalcLambda (a decade for the default -lmin 1e-5 -lmax 1e5 -region 10, plus a
//      margin either side for Newton steps that leave the bracket) a CHEB_N = 24 term series reproduces S to ~6e-15, dS/dt
//      to ~3e-13 and d2S/dt2 to ~6e-11 of its scale (tests/test_cheb_search.py).  The coefficients are LINEAR in the
//      products a_i b_i: a_k = sum_i a_i b_i c_k(delta_i), i.e. one skinny matrix product per SNP batch (lmm_grid.hip.h)
//      gives each SNP the series of its x-dependent sums on the interval its bracket lies in, and the root finder's
//      ~10 evaluations per SNP become O(CHEB_N) work instead of O(n) streaming passes.
//  (3) The derivative form of LogRL_dev1/dev2 and LogL_dev1/dev2 (src/lmm.cpp:544-640,719-797,866-943,1035-1125):
//      with PP = P + dP/dt and PPP = P + 3/2 dP/dt + 1/2 d2P/dt2 (because lambda P K P = P - PP), the reference's
//      differences  P_yy - PP_yy,  P_yy + PPP_yy - 2 PP_yy,  df - trace_P, ...  are  -P_yy',  (P_yy'' - P_yy')/2,
//      g + sum_i w_i'/w_i, ...  (g = sum_i (1 - H_i), w_i the pivots): the same numbers, computed without the
//      cancellation, from the series' own derivatives.
#pragma once
#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define GH_HD __host__ __device__ __forceinline__
#else
#define GH_HD inline
#endif

namespace gemma_hip {

constexpr int CHEB_N = 24; // terms per interval; columns of one x-group in the table product

// ------------------------------------------------------------------ root finders
// GSL roots/brent.c (brent_init / brent_iterate), restated; state lives in registers.
struct Brent {
  double a, b, c, d, e, fa, fb, fc;
  double root, x_lower, x_upper;
};
enum { RS_SUCCESS = 0, RS_CONTINUE = -2, RS_EINVAL = 4, RS_EBADFUNC = 9, RS_EZERODIV = 12, RS_OUTSIDE = 100 };

GH_HD bool finite_d(double v) { return fabs(v) <= DBL_MAX; }

GH_HD int brent_set(Brent &s, double x_lower, double x_upper, double f_lower, double f_upper) {
  // f_lower/f_upper: the reference re-evaluates dev1 at both ends (gsl_root_fsolver_set ->
  // brent_init); the function is pure, so the grid-scan values are the same numbers.
  if (x_lower > x_upper) return RS_EINVAL;
  s.root = 0.5 * (x_lower + x_upper);
  s.x_lower = x_lower;
  s.x_upper = x_upper;
  if (!finite_d(f_lower)) return RS_EBADFUNC;
  if (!finite_d(f_upper)) return RS_EBADFUNC;
  s.a = x_lower; s.fa = f_lower;
  s.b = x_upper; s.fb = f_upper;
  s.c = x_upper; s.fc = f_upper;
  s.d = x_upper - x_lower;
  s.e = x_upper - x_lower;
  if ((f_lower < 0.0 && f_upper < 0.0) || (f_lower > 0.0 && f_upper > 0.0)) return RS_EINVAL;
  return RS_SUCCESS;
}

template <class E>
GH_HD int brent_iterate(Brent &s, E &ev) {
  double tol, m;
  bool ac_equal = false;
  double a = s.a, b = s.b, c = s.c, fa = s.fa, fb = s.fb, fc = s.fc, d = s.d, e = s.e;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) {
    ac_equal = true;
    c = a; fc = fa; d = b - a; e = b - a;
  }
  if (fabs(fc) < fabs(fb)) {
    ac_equal = true;
    a = b; b = c; c = a;
    fa = fb; fb = fc; fc = fa;
  }
  tol = 0.5 * DBL_EPSILON * fabs(b);
  m = 0.5 * (c - b);
  if (fb == 0) {
    s.root = b; s.x_lower = b; s.x_upper = b;
    return RS_SUCCESS;
  }
  if (fabs(m) <= tol) {
    s.root = b;
    if (b < c) { s.x_lower = b; s.x_upper = c; } else { s.x_lower = c; s.x_upper = b; }
    return RS_SUCCESS;
  }
  if (fabs(e) < tol || fabs(fa) <= fabs(fb)) {
    d = m; e = m;
  } else {
    double p, q, r;
    const double sv = fb / fa;
    if (ac_equal) {
      p = 2 * m * sv;
      q = 1 - sv;
    } else {
      q = fa / fc;
      r = fb / fc;
      p = sv * (2 * m * q * (q - r) - (b - a) * (r - 1));
      q = (q - 1) * (r - 1) * (sv - 1);
    }
    if (p > 0) q = -q; else p = -p;
    const double lim1 = 3 * m * q - fabs(tol * q), lim2 = fabs(e * q);
    if (2 * p < (lim1 < lim2 ? lim1 : lim2)) {
      e = d; d = p / q;
    } else {
      d = m; e = m;
    }
  }
  a = b; fa = fb;
  if (fabs(d) > tol) b += d; else b += (m > 0 ? +tol : -tol);
  if (!ev.dev1(b, fb)) return RS_OUTSIDE;
  if (!finite_d(fb)) return RS_EBADFUNC;
  s.a = a; s.b = b; s.c = c; s.d = d; s.e = e; s.fa = fa; s.fb = fb; s.fc = fc;
  s.root = b;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) c = a;
  if (b < c) { s.x_lower = b; s.x_upper = c; } else { s.x_lower = c; s.x_upper = b; }
  return RS_SUCCESS;
}

// gsl_root_test_interval(lo, hi, 0, 1e-1) and gsl_root_test_delta(x1, x0, 0, 1e-5) (GSL
// roots/convergence.c) as used at src/lmm.cpp:2050,2073
GH_HD int test_interval_dev(double lo, double hi, double epsrel) {
  if (lo > hi) return RS_EINVAL;
  double min_abs;
  if ((lo > 0.0 && hi > 0.0) || (lo < 0.0 && hi < 0.0))
    min_abs = fmin(fabs(lo), fabs(hi));
  else
    min_abs = 0;
  return (fabs(hi - lo) < epsrel * min_abs) ? RS_SUCCESS : RS_CONTINUE;
}
GH_HD int test_delta_dev(double x1, double x0, double epsrel) {
  return (fabs(x1 - x0) < epsrel * fabs(x1) || x1 == x0) ? RS_SUCCESS : RS_CONTINUE;
}

// One bracket of CalcLambda (src/lmm.cpp:2024-2102): Brent until the interval is within 10 %, then Newton until two
// iterates agree to 1e-5.  `l` and `l_temp` are CalcLambda's function-scope variables (a Brent error in its first
// iteration leaves `l` at the previous bracket's value, :2040-2047) and are passed through.
enum { PB_OK = 0,      // l = the iterate the reference reports (the one BEFORE the last, :2096), clamped to [l_min, l_max]
       PB_STOP = 1,    // Brent hit max_iter: the reference leaves the bracket loop (:2057-2060)
       PB_FAILED = 2,  // Newton failed / left (l_min, l_max) without converging: lambda = logf = NaN (:2087-2094)
       PB_OUTSIDE = 3  // the evaluator could not evaluate (table left): nothing decided, repeat with another evaluator
};
template <class E>
GH_HD int polish_bracket(E &ev, double lambda_l0, double lambda_h0, double d_lo, double d_hi, double l_min, double l_max,
                         double &l, double &l_temp) {
  Brent bs;
  bs.a = bs.b = bs.c = bs.d = bs.e = bs.fa = bs.fb = bs.fc = 0.0;
  bs.root = bs.x_lower = bs.x_upper = 0.0;
  (void)brent_set(bs, lambda_l0, lambda_h0, d_lo, d_hi);
  int status;
  int iter = 0;
  double lambda_l, lambda_h;
  do {
    iter++;
    status = brent_iterate(bs, ev);
    if (status == RS_OUTSIDE) return PB_OUTSIDE;
    if (status != RS_SUCCESS && status != RS_CONTINUE) break;
    l = bs.root;
    lambda_l = bs.x_lower;
    lambda_h = bs.x_upper;
    status = test_interval_dev(lambda_l, lambda_h, 1e-1);
    if (status != RS_SUCCESS && status != RS_CONTINUE) break;
  } while (status == RS_CONTINUE && iter < 100);
  if (status == RS_CONTINUE) return PB_STOP;
  // Newton, GSL roots/newton.c: set() evaluates (f, df) at the start
  int iter2 = 0;
  double root = l, nf, ndf;
  if (!ev.dev12(root, nf, ndf)) return PB_OUTSIDE;
  do {
    iter2++;
    if (ndf == 0.0) {
      status = RS_EZERODIV;
    } else {
      const double root_new = root - (nf / ndf);
      root = root_new;
      if (!ev.dev12(root_new, nf, ndf)) return PB_OUTSIDE;
      status = (!finite_d(nf) || !finite_d(ndf)) ? RS_EBADFUNC : RS_SUCCESS;
    }
    if (status != RS_SUCCESS && status != RS_CONTINUE) break;
    l_temp = l;
    l = root;
    status = test_delta_dev(l, l_temp, 1e-5);
  } while (status == RS_CONTINUE && iter2 < 100 && l > l_min && l < l_max);
  if (status != RS_SUCCESS) return PB_FAILED;
  l = l_temp; // :2096 -- the previous Newton iterate is reported
  if (l < l_min) l = l_min;
  if (l > l_max) l = l_max;
  return PB_OK;
}

// ------------------------------------------------------------------ Chebyshev series
// Interval j of the lambda grid, widened by `margin` of its length either side: t in [mid - half, mid + half]
struct ChebInterval {
  double mid, half;
};
GH_HD ChebInterval cheb_interval(double lam_lo, double lam_hi, double margin) {
  const double a = log(lam_lo), b = log(lam_hi);
  ChebInterval iv;
  iv.mid = 0.5 * (a + b);
  iv.half = 0.5 * (b - a) * (1.0 + 2.0 * margin);
  return iv;
}
// node m of CHEB_N (roots of T_N, first kind): t_m = mid + half cos(pi (m + 1/2) / N)
GH_HD double cheb_node(const ChebInterval &iv, int m) { return iv.mid + iv.half * cos(M_PI * (m + 0.5) / CHEB_N); }

// Coefficients c_k of the interpolant of f through the CHEB_N nodes: c_k = (2/N) sum_m f(t_m) cos(k pi (m + 1/2) / N),
// c_0 halved, so that f(t) ~ sum_k c_k T_k((t - mid) / half).  fm: the N node values; out: the N coefficients.
GH_HD void cheb_fit(const double *fm, double *out) {
  for (int k = 0; k < CHEB_N; ++k) {
    double s = 0.0;
    for (int m = 0; m < CHEB_N; ++m) s += fm[m] * cos(M_PI * k * (m + 0.5) / CHEB_N);
    out[k] = s * (k == 0 ? 1.0 : 2.0) / CHEB_N;
  }
}

// p(s) = sum_k a_k T_k(s) with its first two derivatives in s (Clenshaw and its derivatives); stride between terms
template <int ORDER>
GH_HD void cheb_eval(const double *__restrict__ a, long stride, double s, double &p0, double &p1, double &p2) {
  double b1 = 0.0, b2 = 0.0, d1 = 0.0, d2 = 0.0, e1 = 0.0, e2 = 0.0;
  const double s2 = 2.0 * s;
#pragma unroll
  for (int k = CHEB_N - 1; k >= 1; --k) {
    if (ORDER >= 2) {
      const double e0 = 4.0 * d1 + s2 * e1 - e2;
      e2 = e1; e1 = e0;
    }
    if (ORDER >= 1) {
      const double d0 = 2.0 * b1 + s2 * d1 - d2;
      d2 = d1; d1 = d0;
    }
    const double b0 = a[k * stride] + s2 * b1 - b2;
    b2 = b1; b1 = b0;
  }
  p0 = a[0] + s * b1 - b2;
  p1 = (ORDER >= 1) ? b1 + s * d1 - d2 : 0.0;
  p2 = (ORDER >= 2) ? 2.0 * d1 + s * e1 - e2 : 0.0;
}

// GetabIndex, GEMMA src/param.cpp:1400-1415 (1-based, symmetric)
template <int C>
GH_HD constexpr int ab_index(int a, int b) {
  return (2 * (C + 2) - ((b <= a) ? b : a) + 2) * (((b <= a) ? b : a) - 1) / 2 + ((b <= a) ? a : b) - ((b <= a) ? b : a);
}

// What one SNP's table-driven evaluations read.  Layouts (written by lmm_grid.hip.h / the host harness):
//   snp : this SNP's series from the interval's table product, element (col) at snp[col * sstride]:
//         col k = series of sum x^2 H (k < CHEB_N), col xa0 + a * CHEB_N + k = series of sum x u_a H
//         (a < C: U^T W column a, a = C: U^T y).  On the device the table is stored column-major over the
//         interval's slots (sstride = slots per interval) so that one thread per SNP reads coalesced.
//   fix : the interval's SNP-independent series, [pair * CHEB_N + k] for the pairs (a <= b) among (w_1..w_C, y) in
//         row-major upper-triangle order, then [npairs * CHEB_N + k] = series of g(t) = sum_i (1 - H_i), [(npairs + 1) ..]
//         = series of log|H| (used by the final pass only), [(npairs + 2) ..] = series of sum_i (1 - H_i)^2
template <int C>
struct ChebSnp {
  const double *snp;
  long sstride;
  const double *fix;
  int xa0;
  double mid, inv_half; // t -> s = (t - mid) * inv_half
  double n;             // individuals
  // "Q form", the intervals below lambda = 1e-3: there S_ab(t) = S0_ab - lambda Q_ab(t), S0_ab = sum_i a_i b_i and
  // Q_ab = sum_i a_i b_i delta_i H_i; dS/dt = -lambda (Q + Q') is O(lambda) S, and a series of S itself good to 1e-15 of S
  // would carry 1e-13 / lambda of relative error in it.  So for those intervals `snp` and the pair part of `fix` hold the
  // series of Q (same table product, weights c_k of delta_i H_i instead of H_i) and the constants come beside them:
  int qform;
  double s0x[C + 2];    // sum_i x_i^2, sum_i x_i u_a(i) (a < C: U^T W column a, a = C: U^T y)
  const double *s0f;    // the SNP-independent pairs, in the order of `fix`
};

// dev1 (ORDER 1) or dev1 and dev2 (ORDER 2) of logRL (REML) / logL at lambda = l from the series; false when l lies
// outside the interval.  Formulas: header comment (3); the projection is the derivative of the Schur recursion of
// CalcPab (src/lmm.cpp:326-349) carried along with it.
template <int C, bool REML, int ORDER>
GH_HD bool cheb_deriv(const ChebSnp<C> &cs, double l, double &dev1, double &dev2) {
  constexpr int NV = C + 2, NI = (C + 3) * (C + 2) / 2, NPAIR = (C + 1) * (C + 2) / 2;
  const double t = log(l);
  const double s = (t - cs.mid) * cs.inv_half;
  if (!(fabs(s) <= 1.0)) return false;
  const double k1 = cs.inv_half, k2 = cs.inv_half * cs.inv_half;
  double p0[NI], p1[NI], p2[NI];
#pragma unroll
  for (int a = 1; a <= NV; ++a) {
#pragma unroll
    for (int b = a; b <= NV; ++b) {
      const int q = ab_index<C>(a, b);
      // variables 1..C: covariates (fixed index a - 1), C + 1: x, C + 2: y (fixed index C)
      const bool ax = (a == C + 1), bx = (b == C + 1);
      const int fa = (a == C + 2) ? C : a - 1, fb = (b == C + 2) ? C : b - 1;
      double v0, v1, v2;
      if (ax && bx) {
        cheb_eval<ORDER>(cs.snp, cs.sstride, s, v0, v1, v2);
      } else if (ax || bx) {
        cheb_eval<ORDER>(cs.snp + (long)(cs.xa0 + (ax ? fb : fa) * CHEB_N) * cs.sstride, cs.sstride, s, v0, v1, v2);
      } else {
        cheb_eval<ORDER>(cs.fix + (fa * (C + 1) - fa * (fa - 1) / 2 + (fb - fa)) * CHEB_N, 1, s, v0, v1, v2);
      }
      if (cs.qform) {
        const double S0 = (ax && bx) ? cs.s0x[0]
                          : (ax || bx) ? cs.s0x[1 + (ax ? fb : fa)]
                                       : cs.s0f[fa * (C + 1) - fa * (fa - 1) / 2 + (fb - fa)];
        const double Q1 = v1 * k1, Q2 = v2 * k2;
        p0[q] = S0 - l * v0;
        p1[q] = -l * (v0 + Q1);
        p2[q] = -l * (v0 + 2.0 * Q1 + Q2);
      } else {
        p0[q] = v0;
        p1[q] = v1 * k1;
        p2[q] = v2 * k2;
      }
    }
  }
  // g = sum_i (1 - H_i) and, for the second derivative, gg = g - dg/dt = sum_i (1 - H_i)^2 from a series of its own (the
  // difference of the two O(lambda) terms cancels to O(lambda^2))
  double g0, g1, g2, gg = 0.0;
  cheb_eval<0>(cs.fix + NPAIR * CHEB_N, 1, s, g0, g1, g2);
  if (ORDER >= 2) cheb_eval<0>(cs.fix + (NPAIR + 2) * CHEB_N, 1, s, gg, g1, g2);
  double sr = 0.0, sq = 0.0;
#pragma unroll
  for (int p = 1; p <= C + 1; ++p) {
    const int iww = ab_index<C>(p, p);
    const double W0 = p0[iww], W1 = p1[iww], W2 = p2[iww];
    const double r = W1 / W0;
    sr += r;
    if (ORDER >= 2) sq += r + r * r - W2 / W0;
    if (W0 != 0) {
      // in place: entries (a, b > p) read only entries with index p of this level, which this level does not write
#pragma unroll
      for (int a = p + 1; a <= NV; ++a) {
#pragma unroll
        for (int b = a; b <= NV; ++b) {
          const int iab = ab_index<C>(a, b), iaw = ab_index<C>(a, p), ibw = ab_index<C>(b, p);
          const double A0 = p0[iaw], A1 = p1[iaw], B0 = p0[ibw], B1 = p1[ibw];
          const double m0 = A0 * B0, m1 = A1 * B0 + A0 * B1;
          const double u0 = m0 / W0;
          const double u1 = (m1 - u0 * W1) / W0;
          if (ORDER >= 2) {
            const double m2 = p2[iaw] * B0 + 2.0 * A1 * B1 + A0 * p2[ibw];
            p2[iab] -= (m2 - 2.0 * u1 * W1 - u0 * W2) / W0;
          }
          p0[iab] -= u0;
          p1[iab] -= u1;
        }
      }
    }
  }
  constexpr int iyy = ab_index<C>(C + 2, C + 2);
  const double Pyy = p0[iyy], Pyy1 = p1[iyy], Pyy2 = p2[iyy];
  const double yPKPy = -Pyy1 / l;
  const double yPKPKPy = 0.5 * (Pyy2 - Pyy1) / (l * l);
  if (REML) {
    const double df = cs.n - (double)C - 1.0;
    const double trace_PK = (g0 + sr) / l;
    dev1 = -0.5 * trace_PK + 0.5 * df * yPKPy / Pyy;
    if (ORDER >= 2) {
      const double trace_PKPK = (gg + sq) / (l * l);
      dev2 = 0.5 * trace_PKPK - 0.5 * df * (2.0 * yPKPKPy * Pyy - yPKPy * yPKPy) / (Pyy * Pyy);
    }
  } else {
    const double trace_HiK = g0 / l;
    dev1 = -0.5 * trace_HiK + 0.5 * cs.n * yPKPy / Pyy;
    if (ORDER >= 2) {
      const double trace_HiKHiK = gg / (l * l);
      dev2 = 0.5 * trace_HiKHiK - 0.5 * cs.n * (2.0 * yPKPKPy * Pyy - yPKPy * yPKPy) / (Pyy * Pyy);
    }
  }
  return true;
}

// The evaluator polish_bracket takes, over one SNP's series on one interval.  A non-finite value counts as "cannot
// evaluate": the reference's handling of those (GSL_EBADFUNC, which can leave CalcLambda's `l` at the previous bracket's
// value, src/lmm.cpp:2040-2047) is reproduced by the streaming evaluator, which then repeats the bracket.
template <int C, bool REML>
struct ChebEvaluator {
  ChebSnp<C> cs;
  GH_HD bool dev1(double l, double &d1) {
    double d2;
    return cheb_deriv<C, REML, 1>(cs, l, d1, d2) && finite_d(d1);
  }
  GH_HD bool dev12(double l, double &d1, double &d2) {
    return cheb_deriv<C, REML, 2>(cs, l, d1, d2) && finite_d(d1) && finite_d(d2);
  }
};

} // namespace gemma_hip
