/* BLAS/LAPACK bridge: TEST INFRASTRUCTURE ONLY (see gsl/gsl_shim_core.h).
 * The reference calls cblas_dgemm (src/fastblas.cpp:139,202), the Fortran LAPACK entry points dgemm_/dpotrf_/dpotrs_/
 * dsyev_/dsyevr_/ddot_ (src/lapack.cpp:33-46) and openblas_get_* (src/gemma.cpp:29-36) by their plain names.  The only
 * OpenBLAS in this image is the one inside scipy, whose symbols carry a scipy_ prefix; these stubs forward to it. */
extern void scipy_cblas_dgemm(int, int, int, int, int, int, double, const double *, int, const double *, int, double, double *, int);
extern void scipy_dgemm_(char *, char *, int *, int *, int *, double *, double *, int *, double *, int *, double *, double *, int *);
extern void scipy_dpotrf_(char *, int *, double *, int *, int *);
extern void scipy_dpotrs_(char *, int *, int *, double *, int *, double *, int *, int *);
extern void scipy_dsyev_(char *, char *, int *, double *, int *, double *, double *, int *, int *);
extern void scipy_dsyevr_(char *, char *, char *, int *, double *, int *, double *, double *, int *, int *, double *, int *,
                          double *, double *, int *, int *, double *, int *, int *, int *, int *);
extern double scipy_ddot_(int *, double *, int *, double *, int *);
extern int scipy_openblas_get_num_threads(void);
extern int scipy_openblas_get_parallel(void);
extern char *scipy_openblas_get_config(void);
extern char *scipy_openblas_get_corename(void);

void cblas_dgemm(int o, int ta, int tb, int m, int n, int k, double alpha, const double *a, int lda, const double *b,
                 int ldb, double beta, double *c, int ldc) {
  scipy_cblas_dgemm(o, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
void dgemm_(char *ta, char *tb, int *m, int *n, int *k, double *alpha, double *a, int *lda, double *b, int *ldb,
            double *beta, double *c, int *ldc) { scipy_dgemm_(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); }
void dpotrf_(char *u, int *n, double *a, int *lda, int *info) { scipy_dpotrf_(u, n, a, lda, info); }
void dpotrs_(char *u, int *n, int *nrhs, double *a, int *lda, double *b, int *ldb, int *info) { scipy_dpotrs_(u, n, nrhs, a, lda, b, ldb, info); }
void dsyev_(char *j, char *u, int *n, double *a, int *lda, double *w, double *work, int *lwork, int *info) {
  scipy_dsyev_(j, u, n, a, lda, w, work, lwork, info);
}
void dsyevr_(char *j, char *r, char *u, int *n, double *a, int *lda, double *vl, double *vu, int *il, int *iu,
             double *abstol, int *m, double *w, double *z, int *ldz, int *isuppz, double *work, int *lwork, int *iwork,
             int *liwork, int *info) {
  scipy_dsyevr_(j, r, u, n, a, lda, vl, vu, il, iu, abstol, m, w, z, ldz, isuppz, work, lwork, iwork, liwork, info);
}
double ddot_(int *n, double *x, int *incx, double *y, int *incy) { return scipy_ddot_(n, x, incx, y, incy); }
int openblas_get_num_threads(void) { return scipy_openblas_get_num_threads(); }
int openblas_get_parallel(void) { return scipy_openblas_get_parallel(); }
char *openblas_get_config(void) { return scipy_openblas_get_config(); }
char *openblas_get_corename(void) { return scipy_openblas_get_corename(); }
