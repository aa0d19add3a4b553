/* GSL-API shim: TEST INFRASTRUCTURE ONLY (part of oracle/, never linked into the product).
 *
 * The reference (GEMMA, /root/reference/src) includes gsl/gsl_*.h and links -lgsl -lopenblas
 * (/root/reference/Makefile:163-165).  GSL is not installed in this image and there is no network, so the
 * reference's own sources cannot be compiled as they are.  This directory supplies the subset of the GSL 2.x C API
 * those sources use -- same type layouts, same function names and argument meaning, GSL's published algorithms
 * restated from its documentation (Brent / Newton root solvers, LU, the F and chi-square tails, MT19937) -- so that
 * `oracle/Makefile ref` can compile /root/reference/src/*.cpp UNCHANGED, where they lie, into oracle/_ref/gemma.
 * BLAS/LAPACK calls go to the OpenBLAS that ships inside scipy (symbol prefix scipy_), the same dgemm / dsyevr
 * routines the reference calls.  Routines that only BSLMM / VC / logistic modes need (multiroot, QR, the random
 * variate generators) are declared and abort when called: they are not on the kinship + LMM path.
 *
 * What this does and does not pin: every line of the reference's own arithmetic (src/lmm.cpp, src/mvlmm.cpp,
 * src/gemma_io.cpp, src/param.cpp, src/mathfunc.cpp, src/lapack.cpp ...) is the reference's; the GSL routines under
 * it are restatements, validated by reproducing the reference's golden values (test/dev_tests.rb) with the binary.
 */
#ifndef GSL_SHIM_CORE_H
#define GSL_SHIM_CORE_H

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- version ---- */
#define GSL_VERSION "2.7-shim"
#define GSL_MAJOR_VERSION 2
#define GSL_MINOR_VERSION 7
extern const char *gsl_version;

/* ---- errno ---- */
enum {
  GSL_SUCCESS = 0, GSL_FAILURE = -1, GSL_CONTINUE = -2, GSL_EDOM = 1, GSL_ERANGE = 2, GSL_EFAULT = 3, GSL_EINVAL = 4,
  GSL_EFAILED = 5, GSL_EFACTOR = 6, GSL_ESANITY = 7, GSL_ENOMEM = 8, GSL_EBADFUNC = 9, GSL_ERUNAWAY = 10,
  GSL_EMAXITER = 11, GSL_EZERODIV = 12, GSL_EBADTOL = 13, GSL_ETOL = 14, GSL_EUNDRFLW = 15, GSL_EOVRFLW = 16,
  GSL_ELOSS = 17, GSL_EROUND = 18, GSL_EBADLEN = 19, GSL_ENOTSQR = 20, GSL_ESING = 21, GSL_EDIVERGE = 22,
  GSL_EUNSUP = 23, GSL_EUNIMPL = 24, GSL_ECACHE = 25, GSL_ETABLE = 26, GSL_ENOPROG = 27, GSL_ENOPROGJ = 28,
  GSL_ETOLF = 29, GSL_ETOLX = 30, GSL_ETOLG = 31, GSL_EOF = 32
};
typedef void gsl_error_handler_t(const char *reason, const char *file, int line, int gsl_errno);
void gsl_error(const char *reason, const char *file, int line, int gsl_errno);
gsl_error_handler_t *gsl_set_error_handler(gsl_error_handler_t *new_handler);
gsl_error_handler_t *gsl_set_error_handler_off(void);
const char *gsl_strerror(const int gsl_errno);
#define GSL_ERROR(reason, gsl_errno) do { gsl_error(reason, __FILE__, __LINE__, gsl_errno); return gsl_errno; } while (0)
#define GSL_ERROR_VAL(reason, gsl_errno, value) do { gsl_error(reason, __FILE__, __LINE__, gsl_errno); return value; } while (0)

/* ---- sys / math ---- */
#define GSL_DBL_EPSILON 2.2204460492503131e-16
#define GSL_DBL_MIN 2.2250738585072014e-308
#define GSL_DBL_MAX 1.7976931348623157e+308
#define GSL_NAN (NAN)
#define GSL_POSINF (INFINITY)
#define GSL_NEGINF (-INFINITY)
#define GSL_MIN(a, b) ((a) < (b) ? (a) : (b))
#define GSL_MAX(a, b) ((a) > (b) ? (a) : (b))
int gsl_isnan(const double x);
int gsl_isinf(const double x);
int gsl_finite(const double x);

typedef struct { double (*function)(double x, void *params); void *params; } gsl_function;
typedef struct {
  double (*f)(double x, void *params);
  double (*df)(double x, void *params);
  void (*fdf)(double x, void *params, double *f, double *df);
  void *params;
} gsl_function_fdf;
#define GSL_FN_EVAL(F, x) (*((F)->function))(x, (F)->params)
#define GSL_FN_FDF_EVAL_F(FDF, x) (*((FDF)->f))(x, (FDF)->params)
#define GSL_FN_FDF_EVAL_DF(FDF, x) (*((FDF)->df))(x, (FDF)->params)
#define GSL_FN_FDF_EVAL_F_DF(FDF, x, y, dy) (*((FDF)->fdf))(x, (FDF)->params, (y), (dy))

/* ---- block / vector / matrix ---- */
typedef struct { size_t size; double *data; } gsl_block;
typedef struct { size_t size; size_t stride; double *data; gsl_block *block; int owner; } gsl_vector;
typedef struct { gsl_vector vector; } _gsl_vector_view;
typedef _gsl_vector_view gsl_vector_view;
typedef struct { gsl_vector vector; } _gsl_vector_const_view;
typedef const _gsl_vector_const_view gsl_vector_const_view;
typedef struct { size_t size1; size_t size2; size_t tda; double *data; gsl_block *block; int owner; } gsl_matrix;
typedef struct { gsl_matrix matrix; } _gsl_matrix_view;
typedef _gsl_matrix_view gsl_matrix_view;
typedef struct { gsl_matrix matrix; } _gsl_matrix_const_view;
typedef const _gsl_matrix_const_view gsl_matrix_const_view;

typedef struct { size_t size; int *data; } gsl_block_int;
typedef struct { size_t size; size_t stride; int *data; gsl_block_int *block; int owner; } gsl_vector_int;
typedef struct { size_t size1; size_t size2; size_t tda; int *data; gsl_block_int *block; int owner; } gsl_matrix_int;

gsl_vector *gsl_vector_alloc(const size_t n);
gsl_vector *gsl_vector_calloc(const size_t n);
void gsl_vector_free(gsl_vector *v);
static inline double gsl_vector_get(const gsl_vector *v, const size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_set(gsl_vector *v, const size_t i, double x) { v->data[i * v->stride] = x; }
static inline double *gsl_vector_ptr(gsl_vector *v, const size_t i) { return v->data + i * v->stride; }
void gsl_vector_set_all(gsl_vector *v, double x);
void gsl_vector_set_zero(gsl_vector *v);
int gsl_vector_memcpy(gsl_vector *dest, const gsl_vector *src);
int gsl_vector_add(gsl_vector *a, const gsl_vector *b);
int gsl_vector_sub(gsl_vector *a, const gsl_vector *b);
int gsl_vector_mul(gsl_vector *a, const gsl_vector *b);
int gsl_vector_div(gsl_vector *a, const gsl_vector *b);
int gsl_vector_scale(gsl_vector *a, const double x);
int gsl_vector_add_constant(gsl_vector *a, const double x);
void gsl_vector_minmax(const gsl_vector *v, double *min_out, double *max_out);
_gsl_vector_view gsl_vector_subvector(gsl_vector *v, size_t i, size_t n);
_gsl_vector_const_view gsl_vector_const_subvector(const gsl_vector *v, size_t i, size_t n);
_gsl_vector_view gsl_vector_view_array(double *v, size_t n);
_gsl_vector_const_view gsl_vector_const_view_array(const double *v, size_t n);

gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2);
gsl_matrix *gsl_matrix_calloc(const size_t n1, const size_t n2);
void gsl_matrix_free(gsl_matrix *m);
static inline double gsl_matrix_get(const gsl_matrix *m, const size_t i, const size_t j) { return m->data[i * m->tda + j]; }
static inline void gsl_matrix_set(gsl_matrix *m, const size_t i, const size_t j, const double x) { m->data[i * m->tda + j] = x; }
static inline double *gsl_matrix_ptr(gsl_matrix *m, const size_t i, const size_t j) { return m->data + (i * m->tda + j); }
void gsl_matrix_set_all(gsl_matrix *m, double x);
void gsl_matrix_set_zero(gsl_matrix *m);
void gsl_matrix_set_identity(gsl_matrix *m);
int gsl_matrix_memcpy(gsl_matrix *dest, const gsl_matrix *src);
int gsl_matrix_add(gsl_matrix *a, const gsl_matrix *b);
int gsl_matrix_sub(gsl_matrix *a, const gsl_matrix *b);
int gsl_matrix_mul_elements(gsl_matrix *a, const gsl_matrix *b);
int gsl_matrix_scale(gsl_matrix *a, const double x);
int gsl_matrix_transpose(gsl_matrix *m);
int gsl_matrix_transpose_memcpy(gsl_matrix *dest, const gsl_matrix *src);
int gsl_matrix_equal(const gsl_matrix *a, const gsl_matrix *b);
int gsl_matrix_set_row(gsl_matrix *m, const size_t i, const gsl_vector *v);
int gsl_matrix_set_col(gsl_matrix *m, const size_t j, const gsl_vector *v);
int gsl_matrix_get_row(gsl_vector *v, const gsl_matrix *m, const size_t i);
int gsl_matrix_get_col(gsl_vector *v, const gsl_matrix *m, const size_t j);
_gsl_vector_view gsl_matrix_row(gsl_matrix *m, const size_t i);
_gsl_vector_view gsl_matrix_column(gsl_matrix *m, const size_t j);
_gsl_vector_view gsl_matrix_diagonal(gsl_matrix *m);
_gsl_vector_view gsl_matrix_subrow(gsl_matrix *m, const size_t i, const size_t offset, const size_t n);
_gsl_vector_const_view gsl_matrix_const_row(const gsl_matrix *m, const size_t i);
_gsl_vector_const_view gsl_matrix_const_column(const gsl_matrix *m, const size_t j);
_gsl_vector_const_view gsl_matrix_const_subrow(const gsl_matrix *m, const size_t i, const size_t offset, const size_t n);
_gsl_matrix_view gsl_matrix_submatrix(gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2);
_gsl_matrix_const_view gsl_matrix_const_submatrix(const gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2);
_gsl_matrix_view gsl_matrix_view_array(double *base, const size_t n1, const size_t n2);
_gsl_matrix_const_view gsl_matrix_const_view_array(const double *base, const size_t n1, const size_t n2);

gsl_vector_int *gsl_vector_int_alloc(const size_t n);
void gsl_vector_int_free(gsl_vector_int *v);
static inline int gsl_vector_int_get(const gsl_vector_int *v, const size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_int_set(gsl_vector_int *v, const size_t i, int x) { v->data[i * v->stride] = x; }
gsl_matrix_int *gsl_matrix_int_alloc(const size_t n1, const size_t n2);
void gsl_matrix_int_free(gsl_matrix_int *m);
static inline int gsl_matrix_int_get(const gsl_matrix_int *m, const size_t i, const size_t j) { return m->data[i * m->tda + j]; }
static inline void gsl_matrix_int_set(gsl_matrix_int *m, const size_t i, const size_t j, const int x) { m->data[i * m->tda + j] = x; }

/* ---- permutation ---- */
typedef struct { size_t size; size_t *data; } gsl_permutation;
gsl_permutation *gsl_permutation_alloc(const size_t n);
gsl_permutation *gsl_permutation_calloc(const size_t n);
void gsl_permutation_init(gsl_permutation *p);
void gsl_permutation_free(gsl_permutation *p);

/* ---- cblas enums (gsl_cblas.h) + the CBLAS / Fortran entry points the reference calls directly ---- */
enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };
enum CBLAS_UPLO { CblasUpper = 121, CblasLower = 122 };
enum CBLAS_DIAG { CblasNonUnit = 131, CblasUnit = 132 };
enum CBLAS_SIDE { CblasLeft = 141, CblasRight = 142 };
typedef enum CBLAS_ORDER CBLAS_ORDER_t;
typedef enum CBLAS_TRANSPOSE CBLAS_TRANSPOSE_t;
typedef enum CBLAS_UPLO CBLAS_UPLO_t;
typedef enum CBLAS_DIAG CBLAS_DIAG_t;
typedef enum CBLAS_SIDE CBLAS_SIDE_t;
#define CBLAS_INDEX size_t
void cblas_dgemm(const enum CBLAS_ORDER Order, const enum CBLAS_TRANSPOSE TransA, const enum CBLAS_TRANSPOSE TransB,
                 const int M, const int N, const int K, const double alpha, const double *A, const int lda,
                 const double *B, const int ldb, const double beta, double *C, const int ldc);

/* ---- gsl_blas ---- */
int gsl_blas_ddot(const gsl_vector *X, const gsl_vector *Y, double *result);
int gsl_blas_daxpy(double alpha, const gsl_vector *X, gsl_vector *Y);
int gsl_blas_dgemv(CBLAS_TRANSPOSE_t TransA, double alpha, const gsl_matrix *A, const gsl_vector *X, double beta, gsl_vector *Y);
int gsl_blas_dger(double alpha, const gsl_vector *X, const gsl_vector *Y, gsl_matrix *A);
int gsl_blas_dsyr(CBLAS_UPLO_t Uplo, double alpha, const gsl_vector *X, gsl_matrix *A);
int gsl_blas_dsyr2(CBLAS_UPLO_t Uplo, double alpha, const gsl_vector *X, const gsl_vector *Y, gsl_matrix *A);
int gsl_blas_dsyrk(CBLAS_UPLO_t Uplo, CBLAS_TRANSPOSE_t Trans, double alpha, const gsl_matrix *A, double beta, gsl_matrix *C);
int gsl_blas_dgemm(CBLAS_TRANSPOSE_t TransA, CBLAS_TRANSPOSE_t TransB, double alpha, const gsl_matrix *A, const gsl_matrix *B, double beta, gsl_matrix *C);
int gsl_blas_dtrsv(CBLAS_UPLO_t Uplo, CBLAS_TRANSPOSE_t TransA, CBLAS_DIAG_t Diag, const gsl_matrix *A, gsl_vector *X);

/* ---- linalg ---- */
int gsl_linalg_LU_decomp(gsl_matrix *A, gsl_permutation *p, int *signum);
int gsl_linalg_LU_solve(const gsl_matrix *LU, const gsl_permutation *p, const gsl_vector *b, gsl_vector *x);
int gsl_linalg_LU_invert(const gsl_matrix *LU, const gsl_permutation *p, gsl_matrix *inverse);
double gsl_linalg_LU_det(gsl_matrix *LU, int signum);
double gsl_linalg_LU_lndet(gsl_matrix *LU);
int gsl_linalg_cholesky_decomp(gsl_matrix *A);
int gsl_linalg_cholesky_decomp1(gsl_matrix *A);
int gsl_linalg_QR_decomp(gsl_matrix *A, gsl_vector *tau);
int gsl_linalg_QR_solve(const gsl_matrix *QR, const gsl_vector *tau, const gsl_vector *b, gsl_vector *x);

/* ---- eigen ---- */
typedef struct { size_t size; double *d; double *sd; } gsl_eigen_symm_workspace;
gsl_eigen_symm_workspace *gsl_eigen_symm_alloc(const size_t n);
void gsl_eigen_symm_free(gsl_eigen_symm_workspace *w);
int gsl_eigen_symm(gsl_matrix *A, gsl_vector *eval, gsl_eigen_symm_workspace *w);

/* ---- cdf / sf ---- */
double gsl_cdf_chisq_Q(const double x, const double nu);
double gsl_cdf_chisq_P(const double x, const double nu);
double gsl_cdf_chisq_Qinv(const double Q, const double nu);
double gsl_cdf_fdist_Q(const double x, const double nu1, const double nu2);
double gsl_cdf_fdist_P(const double x, const double nu1, const double nu2);
double gsl_cdf_gaussian_P(const double x, const double sigma);
double gsl_cdf_gaussian_Q(const double x, const double sigma);
double gsl_sf_exp(const double x);
double gsl_sf_log_1plusx(const double x);

/* ---- roots ---- */
typedef struct {
  const char *name; size_t size;
  int (*set)(void *state, gsl_function *f, double *root, double x_lower, double x_upper);
  int (*iterate)(void *state, gsl_function *f, double *root, double *x_lower, double *x_upper);
} gsl_root_fsolver_type;
typedef struct { const gsl_root_fsolver_type *type; gsl_function *function; double root; double x_lower; double x_upper; void *state; } gsl_root_fsolver;
typedef struct {
  const char *name; size_t size;
  int (*set)(void *state, gsl_function_fdf *f, double *root);
  int (*iterate)(void *state, gsl_function_fdf *f, double *root);
} gsl_root_fdfsolver_type;
typedef struct { const gsl_root_fdfsolver_type *type; gsl_function_fdf *fdf; double root; void *state; } gsl_root_fdfsolver;
gsl_root_fsolver *gsl_root_fsolver_alloc(const gsl_root_fsolver_type *T);
void gsl_root_fsolver_free(gsl_root_fsolver *s);
int gsl_root_fsolver_set(gsl_root_fsolver *s, gsl_function *f, double x_lower, double x_upper);
int gsl_root_fsolver_iterate(gsl_root_fsolver *s);
const char *gsl_root_fsolver_name(const gsl_root_fsolver *s);
double gsl_root_fsolver_root(const gsl_root_fsolver *s);
double gsl_root_fsolver_x_lower(const gsl_root_fsolver *s);
double gsl_root_fsolver_x_upper(const gsl_root_fsolver *s);
gsl_root_fdfsolver *gsl_root_fdfsolver_alloc(const gsl_root_fdfsolver_type *T);
int gsl_root_fdfsolver_set(gsl_root_fdfsolver *s, gsl_function_fdf *fdf, double root);
int gsl_root_fdfsolver_iterate(gsl_root_fdfsolver *s);
void gsl_root_fdfsolver_free(gsl_root_fdfsolver *s);
const char *gsl_root_fdfsolver_name(const gsl_root_fdfsolver *s);
double gsl_root_fdfsolver_root(const gsl_root_fdfsolver *s);
int gsl_root_test_interval(double x_lower, double x_upper, double epsabs, double epsrel);
int gsl_root_test_delta(double x1, double x0, double epsabs, double epsrel);
extern const gsl_root_fsolver_type *gsl_root_fsolver_brent;
extern const gsl_root_fdfsolver_type *gsl_root_fdfsolver_newton;

/* ---- rng / randist ---- */
typedef struct {
  const char *name; unsigned long int max; unsigned long int min; size_t size;
  void (*set)(void *state, unsigned long int seed);
  unsigned long int (*get)(void *state);
  double (*get_double)(void *state);
} gsl_rng_type;
typedef struct { const gsl_rng_type *type; void *state; } gsl_rng;
extern const gsl_rng_type *gsl_rng_default;
extern const gsl_rng_type *gsl_rng_mt19937;
extern unsigned long int gsl_rng_default_seed;
const gsl_rng_type *gsl_rng_env_setup(void);
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T);
void gsl_rng_free(gsl_rng *r);
void gsl_rng_set(const gsl_rng *r, unsigned long int seed);
const char *gsl_rng_name(const gsl_rng *r);
unsigned long int gsl_rng_get(const gsl_rng *r);
double gsl_rng_uniform(const gsl_rng *r);
unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n);
int gsl_ran_choose(const gsl_rng *r, void *dest, size_t k, void *src, size_t n, size_t size);
/* off-path (BSLMM / VC): declared, abort when called */
typedef struct { size_t K; size_t *A; double *F; } gsl_ran_discrete_t;
gsl_ran_discrete_t *gsl_ran_discrete_preproc(size_t K, const double *P);
size_t gsl_ran_discrete(const gsl_rng *r, const gsl_ran_discrete_t *g);
void gsl_ran_discrete_free(gsl_ran_discrete_t *g);
double gsl_ran_gamma(const gsl_rng *r, const double a, const double b);
double gsl_ran_gaussian(const gsl_rng *r, const double sigma);
double gsl_ran_geometric_pdf(const unsigned int k, const double p);

/* ---- multiroots (off-path: VC) ---- */
typedef struct {
  int (*f)(const gsl_vector *x, void *params, gsl_vector *f);
  int (*df)(const gsl_vector *x, void *params, gsl_matrix *df);
  int (*fdf)(const gsl_vector *x, void *params, gsl_vector *f, gsl_matrix *df);
  size_t n; void *params;
} gsl_multiroot_function_fdf;
typedef struct { const char *name; } gsl_multiroot_fdfsolver_type;
typedef struct { const gsl_multiroot_fdfsolver_type *type; gsl_multiroot_function_fdf *fdf; gsl_vector *x; gsl_vector *f; gsl_matrix *J; gsl_vector *dx; void *state; } gsl_multiroot_fdfsolver;
extern const gsl_multiroot_fdfsolver_type *gsl_multiroot_fdfsolver_hybridsj;
gsl_multiroot_fdfsolver *gsl_multiroot_fdfsolver_alloc(const gsl_multiroot_fdfsolver_type *T, size_t n);
void gsl_multiroot_fdfsolver_free(gsl_multiroot_fdfsolver *s);
int gsl_multiroot_fdfsolver_set(gsl_multiroot_fdfsolver *s, gsl_multiroot_function_fdf *fdf, const gsl_vector *x);
int gsl_multiroot_fdfsolver_iterate(gsl_multiroot_fdfsolver *s);
int gsl_multiroot_test_residual(const gsl_vector *f, double epsabs);

#ifdef __cplusplus
}
#endif
#endif
