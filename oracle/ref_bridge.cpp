// Test infrastructure only: C entry points onto the REFERENCE's own free functions (src/mvlmm.cpp), so that a test can
// call them with plain arrays through ctypes and compare function by function with the restatement in mvlmm_oracle.c.
// Linked with the reference's objects (oracle/Makefile `ref` -> oracle/_ref/libgemma_ref.so); nothing is copied: the
// declarations below repeat the signatures the reference defines at the cited lines.
#include <cmath>
#include <cstring>
#include <functional>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include "gsl/gsl_matrix.h"
#include "gsl/gsl_vector.h"
#include "lmm.h"  // the reference's own header (-I/root/reference/src): class LMM, SUMSTAT, SnpNameValues

// src/lapack.h:34 (definition src/lapack.cpp:260-291): all eigenpairs through dsyevr_, eval < 1e-10 -> 0
double EigenDecomp_Zeroed(gsl_matrix *G, gsl_matrix *U, gsl_vector *eval, const size_t flag_largematrix);
// src/gemma_io.h:94 (definition src/gemma_io.cpp:1599-1738): the -gk loop over a .bed file
bool PlinkKin(const std::string &file_bed, std::vector<int> &indicator_snp, const int k_mode, const int display_pace,
              gsl_matrix *matrix_kin);

// src/mvlmm.cpp:599-604
double MphEM(const char func_name, const size_t max_iter, const double max_prec, const gsl_vector *eval, const gsl_matrix *X,
             const gsl_matrix *Y, gsl_matrix *U_hat, gsl_matrix *E_hat, gsl_matrix *OmegaU, gsl_matrix *OmegaE,
             gsl_matrix *UltVehiY, gsl_matrix *UltVehiBX, gsl_matrix *UltVehiU, gsl_matrix *UltVehiE, gsl_matrix *V_g,
             gsl_matrix *V_e, gsl_matrix *B);
// src/mvlmm.cpp:727-729
double MphCalcP(const gsl_vector *eval, const gsl_vector *x_vec, const gsl_matrix *W, const gsl_matrix *Y,
                const gsl_matrix *V_g, const gsl_matrix *V_e, gsl_matrix *UltVehiY, gsl_vector *beta, gsl_matrix *Vbeta);
// src/mvlmm.cpp:2608-2613
double MphNR(const char func_name, const size_t max_iter, const double max_prec, const gsl_vector *eval, const gsl_matrix *X,
             const gsl_matrix *Y, gsl_matrix *Hi_all, gsl_matrix *xHi_all, gsl_matrix *Hiy_all, gsl_matrix *V_g,
             gsl_matrix *V_e, gsl_matrix *Hessian_inv, double &crt_a, double &crt_b, double &crt_c);
// src/mvlmm.cpp:2952-2953
double PCRT(const size_t mode, const size_t d_size, const double p_value, const double crt_a, const double crt_b, const double crt_c);
// src/mvlmm.cpp:213-214
double EigenProc(const gsl_matrix *V_g, const gsl_matrix *V_e, gsl_vector *D_l, gsl_matrix *UltVeh, gsl_matrix *UltVehi);

namespace {
gsl_matrix_view mview(const double *p, size_t r, size_t c) { return gsl_matrix_view_array(const_cast<double *>(p), r, c); }
gsl_vector_view vview(const double *p, size_t n) { return gsl_vector_view_array(const_cast<double *>(p), n); }
}  // namespace

extern "C" {

double ref_MphEM(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval,
                 const double *X /* c x n */, const double *Y /* d x n */, double *Vg, double *Ve, double *B /* d x c */) {
  gsl_vector_view ev = vview(eval, n);
  gsl_matrix_view Xm = mview(X, c, n), Ym = mview(Y, d, n), Vgm = mview(Vg, d, d), Vem = mview(Ve, d, d), Bm = mview(B, d, c);
  gsl_matrix *t[8];
  for (int i = 0; i < 8; i++) t[i] = gsl_matrix_alloc(d, n);
  double l = MphEM(func, max_iter, max_prec, &ev.vector, &Xm.matrix, &Ym.matrix, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7],
                   &Vgm.matrix, &Vem.matrix, &Bm.matrix);
  for (int i = 0; i < 8; i++) gsl_matrix_free(t[i]);
  return l;
}

double ref_MphNR(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval, const double *X,
                 const double *Y, double *Vg, double *Ve, double *Hessian_inv /* d(d+1) x d(d+1) */) {
  gsl_vector_view ev = vview(eval, n);
  gsl_matrix_view Xm = mview(X, c, n), Ym = mview(Y, d, n), Vgm = mview(Vg, d, d), Vem = mview(Ve, d, d);
  gsl_matrix_view Hm = mview(Hessian_inv, d * (d + 1), d * (d + 1));
  gsl_matrix *Hi_all = gsl_matrix_alloc(d, d * n), *xHi_all = gsl_matrix_alloc(d * c, d * n), *Hiy_all = gsl_matrix_alloc(d, n);
  double a, b, cc;
  double l = MphNR(func, max_iter, max_prec, &ev.vector, &Xm.matrix, &Ym.matrix, Hi_all, xHi_all, Hiy_all, &Vgm.matrix, &Vem.matrix,
                   &Hm.matrix, a, b, cc);
  gsl_matrix_free(Hi_all); gsl_matrix_free(xHi_all); gsl_matrix_free(Hiy_all);
  return l;
}

// the same call, also handing back crt_a, crt_b, crt_c of the last CalcDev (src/mvlmm.cpp:2054-2331, :2522-2530)
double ref_MphNR_crt(char func, size_t max_iter, double max_prec, size_t n, size_t d, size_t c, const double *eval, const double *X,
                     const double *Y, double *Vg, double *Ve, double *Hessian_inv, double *crt /* 3 */) {
  gsl_vector_view ev = vview(eval, n);
  gsl_matrix_view Xm = mview(X, c, n), Ym = mview(Y, d, n), Vgm = mview(Vg, d, d), Vem = mview(Ve, d, d);
  gsl_matrix_view Hm = mview(Hessian_inv, d * (d + 1), d * (d + 1));
  gsl_matrix *Hi_all = gsl_matrix_alloc(d, d * n), *xHi_all = gsl_matrix_alloc(d * c, d * n), *Hiy_all = gsl_matrix_alloc(d, n);
  double l = MphNR(func, max_iter, max_prec, &ev.vector, &Xm.matrix, &Ym.matrix, Hi_all, xHi_all, Hiy_all, &Vgm.matrix, &Vem.matrix,
                   &Hm.matrix, crt[0], crt[1], crt[2]);
  gsl_matrix_free(Hi_all); gsl_matrix_free(xHi_all); gsl_matrix_free(Hiy_all);
  return l;
}

double ref_PCRT(size_t mode, size_t d, double p, double a, double b, double c) { return PCRT(mode, d, p, a, b, c); }

double ref_MphCalcP(size_t n, size_t d, size_t cw, const double *eval, const double *x, const double *W /* cw x n */,
                    const double *Y, const double *Vg, const double *Ve, double *beta, double *Vbeta) {
  gsl_vector_view ev = vview(eval, n), xv = vview(x, n), bv = vview(beta, d);
  gsl_matrix_view Wm = mview(W, cw, n), Ym = mview(Y, d, n), Vgm = mview(Vg, d, d), Vem = mview(Ve, d, d), Vb = mview(Vbeta, d, d);
  gsl_matrix *UltVehiY = gsl_matrix_alloc(d, n);
  double p = MphCalcP(&ev.vector, &xv.vector, &Wm.matrix, &Ym.matrix, &Vgm.matrix, &Vem.matrix, UltVehiY, &bv.vector, &Vb.matrix);
  gsl_matrix_free(UltVehiY);
  return p;
}

double ref_EigenProc(size_t d, const double *Vg, const double *Ve, double *Dl, double *UltVeh, double *UltVehi) {
  gsl_matrix_view Vgm = mview(Vg, d, d), Vem = mview(Ve, d, d), A = mview(UltVeh, d, d), Bi = mview(UltVehi, d, d);
  gsl_vector_view D = vview(Dl, d);
  return EigenProc(&Vgm.matrix, &Vem.matrix, &D.vector, &A.matrix, &Bi.matrix);
}

// The reference's univariate driver itself: LMM::Analyze (src/lmm.cpp:1474-1658) -- its batching into Xlarge, mean imputation,
// fast_dgemm("T","N",U,Xlarge) and the per-SNP loop of batch_compute (CalcUab, CalcRLScore, CalcLambda, CalcRLWald, LRT) -- fed
// from memory through the fetch_snp callback AnalyzeBimbam builds from a file (:1675-1700).  X is SNP-major, l x n, NaN = NA;
// every individual is analysed.  out: l records of SUMSTAT (8 doubles, src/param.h:54-66).  Used by tests and by bench.py's
// cpu_baseline leg ("kind": "reference"); returns the number of SNPs the reference produced.
long ref_lmm_analyze(int a_mode, size_t n, size_t c, size_t l, const double *U, const double *eval, const double *UtW,
                     const double *Uty, const double *W, const double *y, const double *X, double l_min, double l_max,
                     size_t n_region, double l_mle_null, double logl_mle_H0, double *out) {
  LMM lmm;
  lmm.a_mode = a_mode;
  lmm.d_pace = 100000000;
  lmm.l_min = l_min; lmm.l_max = l_max; lmm.n_region = n_region;
  lmm.l_mle_null = l_mle_null; lmm.logl_mle_H0 = logl_mle_H0;
  lmm.ni_total = lmm.ni_test = n;
  lmm.ns_total = lmm.ns_test = l;
  lmm.n_cvt = c;
  lmm.time_UtX = lmm.time_opt = 0.0;
  lmm.indicator_idv.assign(n, 1);
  lmm.indicator_snp.assign(l, 1);
  gsl_matrix_view Um = mview(U, n, n), UtWm = mview(UtW, n, c), Wm = mview(W, n, c);
  gsl_vector_view ev = vview(eval, n), Utyv = vview(Uty, n), yv = vview(y, n);
  std::function<SnpNameValues(size_t)> fetch = [&](size_t t) {
    std::vector<double> gs(X + t * n, X + (t + 1) * n);
    return std::make_tuple(std::string("s") + std::to_string(t), gs);
  };
  std::ostringstream sink;  // the progress bar goes to cout
  std::streambuf *old = std::cout.rdbuf(sink.rdbuf());
  lmm.Analyze(fetch, &Um.matrix, &ev.vector, &UtWm.matrix, &Utyv.vector, &Wm.matrix, &yv.vector, std::set<std::string>());
  std::cout.rdbuf(old);
  const size_t m = lmm.sumStat.size() < l ? lmm.sumStat.size() : l;
  for (size_t i = 0; i < m; i++) {
    const SUMSTAT &s = lmm.sumStat[i];
    double *o = out + 8 * i;
    o[0] = s.beta; o[1] = s.se; o[2] = s.lambda_remle; o[3] = s.lambda_mle;
    o[4] = s.p_wald; o[5] = s.p_lrt; o[6] = s.p_score; o[7] = s.logl_H1;
  }
  return (long)lmm.sumStat.size();
}

// The reference's setup stages, for bench.py's cpu_baseline.setup leg (timed on the GPU box's host cores, never part of `value`).
// EigenDecomp_Zeroed (src/lapack.cpp:260-291 -> lapack_eigen_symmv :149-236 -> dsyevr_): G is overwritten as the reference does.
double ref_eigen_decomp_zeroed(size_t n, double *G, double *U, double *eval) {
  gsl_matrix_view Gm = gsl_matrix_view_array(G, n, n), Um = gsl_matrix_view_array(U, n, n);
  gsl_vector_view ev = gsl_vector_view_array(eval, n);
  return EigenDecomp_Zeroed(&Gm.matrix, &Um.matrix, &ev.vector, 0);
}
// PlinkKin (src/gemma_io.cpp:1599-1738) on a .bed file of ns SNPs x ni individuals, every SNP used; K: ni x ni, overwritten
int ref_plink_kin(const char *file_bed, size_t ni, size_t ns, int k_mode, double *K) {
  gsl_matrix_view Km = gsl_matrix_view_array(K, ni, ni);
  std::vector<int> ind(ns, 1);
  std::ostringstream sink;
  std::streambuf *old = std::cout.rdbuf(sink.rdbuf());
  const bool ok = PlinkKin(std::string(file_bed), ind, k_mode, 100000000, &Km.matrix);
  std::cout.rdbuf(old);
  return ok ? 0 : 1;
}
}
