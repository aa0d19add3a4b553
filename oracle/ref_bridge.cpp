// Test infrastructure only: C entry points onto the REFERENCE's own free functions (src/mvlmm.cpp), so that a test can
// call them with plain arrays through ctypes and compare function by function with the restatement in mvlmm_oracle.c.
// Linked with the reference's objects (oracle/Makefile `ref` -> oracle/_ref/libgemma_ref.so); nothing is copied: the
// declarations below repeat the signatures the reference defines at the cited lines.
#include <cstring>
#include "gsl/gsl_matrix.h"
#include "gsl/gsl_vector.h"

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
}
