// GSL-API shim implementation: TEST INFRASTRUCTURE ONLY (see gsl/gsl_shim_core.h).  Written from GSL's documented
// behaviour; BLAS level 1-3 and the symmetric eigenvalue routine forward to the OpenBLAS inside scipy
// (blas_bridge.c), the library the reference itself would call.
#include "gsl/gsl_shim_core.h"

#include <cfloat>
#include <cmath>
#include <cstdint>

extern "C" {
// scipy's OpenBLAS (LP64) exports
double scipy_cblas_ddot(const int n, const double *x, const int incx, const double *y, const int incy);
void scipy_cblas_daxpy(const int n, const double a, const double *x, const int incx, double *y, const int incy);
void scipy_cblas_dgemv(const int order, const int trans, const int m, const int n, const double alpha, const double *a,
                       const int lda, const double *x, const int incx, const double beta, double *y, const int incy);
void scipy_cblas_dger(const int order, const int m, const int n, const double alpha, const double *x, const int incx,
                      const double *y, const int incy, double *a, const int lda);
void scipy_cblas_dsyr(const int order, const int uplo, const int n, const double alpha, const double *x, const int incx,
                      double *a, const int lda);
void scipy_cblas_dsyr2(const int order, const int uplo, const int n, const double alpha, const double *x, const int incx,
                       const double *y, const int incy, double *a, const int lda);
void scipy_cblas_dsyrk(const int order, const int uplo, const int trans, const int n, const int k, const double alpha,
                       const double *a, const int lda, const double beta, double *c, const int ldc);
void scipy_cblas_dgemm(const int order, const int ta, const int tb, const int m, const int n, const int k,
                       const double alpha, const double *a, const int lda, const double *b, const int ldb,
                       const double beta, double *c, const int ldc);
void scipy_cblas_dtrsv(const int order, const int uplo, const int trans, const int diag, const int n, const double *a,
                       const int lda, double *x, const int incx);
void scipy_dsyev_(char *jobz, char *uplo, int *n, double *a, int *lda, double *w, double *work, int *lwork, int *info);
}

static void off_path(const char *what) {
  fprintf(stderr, "gsl shim: %s is not provided (only the kinship + LMM path of the reference is supported)\n", what);
  abort();
}

extern "C" {

const char *gsl_version = GSL_VERSION;

// ---------------------------------------------------------------- errors
static gsl_error_handler_t *g_handler = NULL;
static int g_handler_off = 0;
static void no_handler(const char *, const char *, int, int) {}
void gsl_error(const char *reason, const char *file, int line, int gsl_errno) {
  if (g_handler) { (*g_handler)(reason, file, line, gsl_errno); return; }
  fprintf(stderr, "gsl: %s:%d: ERROR: %s\nDefault GSL error handler invoked.\n", file, line, reason);
  abort();
}
gsl_error_handler_t *gsl_set_error_handler(gsl_error_handler_t *h) {
  gsl_error_handler_t *prev = g_handler;
  g_handler = h;
  return prev;
}
gsl_error_handler_t *gsl_set_error_handler_off(void) {
  gsl_error_handler_t *prev = g_handler;
  g_handler = no_handler;
  (void)g_handler_off;
  return prev;
}
const char *gsl_strerror(const int e) {
  switch (e) {
    case GSL_SUCCESS: return "success";
    case GSL_FAILURE: return "failure";
    case GSL_CONTINUE: return "the iteration has not converged yet";
    case GSL_EDOM: return "input domain error";
    case GSL_ERANGE: return "output range error";
    case GSL_EINVAL: return "invalid argument supplied by user";
    case GSL_EBADFUNC: return "problem with user-supplied function";
    case GSL_EZERODIV: return "tried to divide by zero";
    case GSL_EBADLEN: return "matrix/vector sizes are not conformant";
    case GSL_ENOTSQR: return "matrix not square";
    case GSL_ESING: return "singularity or extremely bad function behavior detected";
    case GSL_ENOMEM: return "malloc failed";
    case GSL_EMAXITER: return "exceeded max number of iterations";
    default: return "unknown error code";
  }
}

int gsl_isnan(const double x) { return std::isnan(x) ? 1 : 0; }
int gsl_isinf(const double x) { return std::isinf(x) ? (x > 0 ? 1 : -1) : 0; }
int gsl_finite(const double x) { return std::isfinite(x) ? 1 : 0; }

// ---------------------------------------------------------------- vector
static gsl_block *block_alloc(size_t n) {
  gsl_block *b = (gsl_block *)malloc(sizeof(gsl_block));
  if (!b) GSL_ERROR_VAL("failed to allocate space for block struct", GSL_ENOMEM, 0);
  b->data = (double *)malloc((n ? n : 1) * sizeof(double));
  if (!b->data) { free(b); GSL_ERROR_VAL("failed to allocate space for block data", GSL_ENOMEM, 0); }
  b->size = n;
  return b;
}
gsl_vector *gsl_vector_alloc(const size_t n) {
  gsl_block *b = block_alloc(n);
  if (!b) return 0;
  gsl_vector *v = (gsl_vector *)malloc(sizeof(gsl_vector));
  v->size = n; v->stride = 1; v->data = b->data; v->block = b; v->owner = 1;
  return v;
}
gsl_vector *gsl_vector_calloc(const size_t n) {
  gsl_vector *v = gsl_vector_alloc(n);
  if (v) memset(v->data, 0, n * sizeof(double));
  return v;
}
void gsl_vector_free(gsl_vector *v) {
  if (!v) return;
  if (v->owner && v->block) { free(v->block->data); free(v->block); }
  free(v);
}
void gsl_vector_set_all(gsl_vector *v, double x) { for (size_t i = 0; i < v->size; i++) v->data[i * v->stride] = x; }
void gsl_vector_set_zero(gsl_vector *v) { gsl_vector_set_all(v, 0.0); }
int gsl_vector_memcpy(gsl_vector *d, const gsl_vector *s) {
  if (d->size != s->size) GSL_ERROR("vector lengths are not equal", GSL_EBADLEN);
  for (size_t i = 0; i < s->size; i++) d->data[i * d->stride] = s->data[i * s->stride];
  return GSL_SUCCESS;
}
#define VEC_OP(name, op)                                                              \
  int name(gsl_vector *a, const gsl_vector *b) {                                      \
    if (a->size != b->size) GSL_ERROR("vectors must have same length", GSL_EBADLEN);  \
    for (size_t i = 0; i < a->size; i++) a->data[i * a->stride] op b->data[i * b->stride]; \
    return GSL_SUCCESS;                                                               \
  }
VEC_OP(gsl_vector_add, +=)
VEC_OP(gsl_vector_sub, -=)
VEC_OP(gsl_vector_mul, *=)
VEC_OP(gsl_vector_div, /=)
int gsl_vector_scale(gsl_vector *a, const double x) { for (size_t i = 0; i < a->size; i++) a->data[i * a->stride] *= x; return GSL_SUCCESS; }
int gsl_vector_add_constant(gsl_vector *a, const double x) { for (size_t i = 0; i < a->size; i++) a->data[i * a->stride] += x; return GSL_SUCCESS; }
void gsl_vector_minmax(const gsl_vector *v, double *mn, double *mx) {
  double lo = v->data[0], hi = v->data[0];
  for (size_t i = 0; i < v->size; i++) {
    double x = v->data[i * v->stride];
    if (x < lo) lo = x;
    if (x > hi) hi = x;
    if (std::isnan(x)) { lo = x; hi = x; break; }
  }
  *mn = lo; *mx = hi;
}
static _gsl_vector_view vview(double *data, size_t n, size_t stride, gsl_block *blk) {
  _gsl_vector_view w;
  w.vector.size = n; w.vector.stride = stride; w.vector.data = data; w.vector.block = blk; w.vector.owner = 0;
  return w;
}
static _gsl_vector_const_view cvview(const double *data, size_t n, size_t stride, gsl_block *blk) {
  _gsl_vector_const_view w;
  w.vector.size = n; w.vector.stride = stride; w.vector.data = (double *)data; w.vector.block = blk; w.vector.owner = 0;
  return w;
}
_gsl_vector_view gsl_vector_subvector(gsl_vector *v, size_t i, size_t n) {
  if (i + (n > 0 ? n - 1 : 0) >= v->size && n > 0) { gsl_error("view would extend past end of vector", __FILE__, __LINE__, GSL_EINVAL); }
  return vview(v->data + i * v->stride, n, v->stride, v->block);
}
_gsl_vector_const_view gsl_vector_const_subvector(const gsl_vector *v, size_t i, size_t n) {
  if (i + (n > 0 ? n - 1 : 0) >= v->size && n > 0) { gsl_error("view would extend past end of vector", __FILE__, __LINE__, GSL_EINVAL); }
  return cvview(v->data + i * v->stride, n, v->stride, v->block);
}
_gsl_vector_view gsl_vector_view_array(double *base, size_t n) { return vview(base, n, 1, 0); }
_gsl_vector_const_view gsl_vector_const_view_array(const double *base, size_t n) { return cvview(base, n, 1, 0); }

// ---------------------------------------------------------------- matrix
gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2) {
  gsl_block *b = block_alloc(n1 * n2);
  if (!b) return 0;
  gsl_matrix *m = (gsl_matrix *)malloc(sizeof(gsl_matrix));
  m->size1 = n1; m->size2 = n2; m->tda = n2; m->data = b->data; m->block = b; m->owner = 1;
  return m;
}
gsl_matrix *gsl_matrix_calloc(const size_t n1, const size_t n2) {
  gsl_matrix *m = gsl_matrix_alloc(n1, n2);
  if (m) memset(m->data, 0, n1 * n2 * sizeof(double));
  return m;
}
void gsl_matrix_free(gsl_matrix *m) {
  if (!m) return;
  if (m->owner && m->block) { free(m->block->data); free(m->block); }
  free(m);
}
void gsl_matrix_set_all(gsl_matrix *m, double x) {
  for (size_t i = 0; i < m->size1; i++) for (size_t j = 0; j < m->size2; j++) m->data[i * m->tda + j] = x;
}
void gsl_matrix_set_zero(gsl_matrix *m) { gsl_matrix_set_all(m, 0.0); }
void gsl_matrix_set_identity(gsl_matrix *m) {
  for (size_t i = 0; i < m->size1; i++) for (size_t j = 0; j < m->size2; j++) m->data[i * m->tda + j] = (i == j) ? 1.0 : 0.0;
}
int gsl_matrix_memcpy(gsl_matrix *d, const gsl_matrix *s) {
  if (d->size1 != s->size1 || d->size2 != s->size2) GSL_ERROR("matrix sizes are different", GSL_EBADLEN);
  for (size_t i = 0; i < s->size1; i++) memmove(d->data + i * d->tda, s->data + i * s->tda, s->size2 * sizeof(double));
  return GSL_SUCCESS;
}
#define MAT_OP(name, op)                                                                          \
  int name(gsl_matrix *a, const gsl_matrix *b) {                                                  \
    if (a->size1 != b->size1 || a->size2 != b->size2) GSL_ERROR("matrices must have same dimensions", GSL_EBADLEN); \
    for (size_t i = 0; i < a->size1; i++)                                                         \
      for (size_t j = 0; j < a->size2; j++) a->data[i * a->tda + j] op b->data[i * b->tda + j];   \
    return GSL_SUCCESS;                                                                           \
  }
MAT_OP(gsl_matrix_add, +=)
MAT_OP(gsl_matrix_sub, -=)
MAT_OP(gsl_matrix_mul_elements, *=)
int gsl_matrix_scale(gsl_matrix *a, const double x) {
  for (size_t i = 0; i < a->size1; i++) for (size_t j = 0; j < a->size2; j++) a->data[i * a->tda + j] *= x;
  return GSL_SUCCESS;
}
int gsl_matrix_transpose(gsl_matrix *m) {
  if (m->size1 != m->size2) GSL_ERROR("matrix must be square to take transpose", GSL_ENOTSQR);
  for (size_t i = 0; i < m->size1; i++)
    for (size_t j = i + 1; j < m->size2; j++) {
      double t = m->data[i * m->tda + j];
      m->data[i * m->tda + j] = m->data[j * m->tda + i];
      m->data[j * m->tda + i] = t;
    }
  return GSL_SUCCESS;
}
int gsl_matrix_transpose_memcpy(gsl_matrix *d, const gsl_matrix *s) {
  if (d->size2 != s->size1 || d->size1 != s->size2) GSL_ERROR("dimensions of dest matrix must be transpose of src matrix", GSL_EBADLEN);
  for (size_t i = 0; i < d->size1; i++) for (size_t j = 0; j < d->size2; j++) d->data[i * d->tda + j] = s->data[j * s->tda + i];
  return GSL_SUCCESS;
}
int gsl_matrix_equal(const gsl_matrix *a, const gsl_matrix *b) {
  if (a->size1 != b->size1 || a->size2 != b->size2) GSL_ERROR_VAL("matrices must have same dimensions", GSL_EBADLEN, 0);
  for (size_t i = 0; i < a->size1; i++) for (size_t j = 0; j < a->size2; j++) if (a->data[i * a->tda + j] != b->data[i * b->tda + j]) return 0;
  return 1;
}
int gsl_matrix_set_row(gsl_matrix *m, const size_t i, const gsl_vector *v) {
  if (i >= m->size1) GSL_ERROR("row index is out of range", GSL_EINVAL);
  if (v->size != m->size2) GSL_ERROR("matrix row size and vector length are not equal", GSL_EBADLEN);
  for (size_t j = 0; j < m->size2; j++) m->data[i * m->tda + j] = v->data[j * v->stride];
  return GSL_SUCCESS;
}
int gsl_matrix_set_col(gsl_matrix *m, const size_t j, const gsl_vector *v) {
  if (j >= m->size2) GSL_ERROR("column index is out of range", GSL_EINVAL);
  if (v->size != m->size1) GSL_ERROR("matrix column size and vector length are not equal", GSL_EBADLEN);
  for (size_t i = 0; i < m->size1; i++) m->data[i * m->tda + j] = v->data[i * v->stride];
  return GSL_SUCCESS;
}
int gsl_matrix_get_row(gsl_vector *v, const gsl_matrix *m, const size_t i) {
  if (i >= m->size1) GSL_ERROR("row index is out of range", GSL_EINVAL);
  if (v->size != m->size2) GSL_ERROR("matrix row size and vector length are not equal", GSL_EBADLEN);
  for (size_t j = 0; j < m->size2; j++) v->data[j * v->stride] = m->data[i * m->tda + j];
  return GSL_SUCCESS;
}
int gsl_matrix_get_col(gsl_vector *v, const gsl_matrix *m, const size_t j) {
  if (j >= m->size2) GSL_ERROR("column index is out of range", GSL_EINVAL);
  if (v->size != m->size1) GSL_ERROR("matrix column size and vector length are not equal", GSL_EBADLEN);
  for (size_t i = 0; i < m->size1; i++) v->data[i * v->stride] = m->data[i * m->tda + j];
  return GSL_SUCCESS;
}
_gsl_vector_view gsl_matrix_row(gsl_matrix *m, const size_t i) {
  if (i >= m->size1) gsl_error("row index is out of range", __FILE__, __LINE__, GSL_EINVAL);
  return vview(m->data + i * m->tda, m->size2, 1, m->block);
}
_gsl_vector_view gsl_matrix_column(gsl_matrix *m, const size_t j) {
  if (j >= m->size2) gsl_error("column index is out of range", __FILE__, __LINE__, GSL_EINVAL);
  return vview(m->data + j, m->size1, m->tda, m->block);
}
_gsl_vector_view gsl_matrix_diagonal(gsl_matrix *m) {
  return vview(m->data, GSL_MIN(m->size1, m->size2), m->tda + 1, m->block);
}
_gsl_vector_view gsl_matrix_subrow(gsl_matrix *m, const size_t i, const size_t off, const size_t n) {
  if (i >= m->size1 || off + n > m->size2) gsl_error("subrow out of range", __FILE__, __LINE__, GSL_EINVAL);
  return vview(m->data + i * m->tda + off, n, 1, m->block);
}
_gsl_vector_const_view gsl_matrix_const_row(const gsl_matrix *m, const size_t i) {
  if (i >= m->size1) gsl_error("row index is out of range", __FILE__, __LINE__, GSL_EINVAL);
  return cvview(m->data + i * m->tda, m->size2, 1, m->block);
}
_gsl_vector_const_view gsl_matrix_const_column(const gsl_matrix *m, const size_t j) {
  if (j >= m->size2) gsl_error("column index is out of range", __FILE__, __LINE__, GSL_EINVAL);
  return cvview(m->data + j, m->size1, m->tda, m->block);
}
_gsl_vector_const_view gsl_matrix_const_subrow(const gsl_matrix *m, const size_t i, const size_t off, const size_t n) {
  if (i >= m->size1 || off + n > m->size2) gsl_error("subrow out of range", __FILE__, __LINE__, GSL_EINVAL);
  return cvview(m->data + i * m->tda + off, n, 1, m->block);
}
_gsl_matrix_view gsl_matrix_submatrix(gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2) {
  if (i + n1 > m->size1 || j + n2 > m->size2) gsl_error("submatrix out of range", __FILE__, __LINE__, GSL_EINVAL);
  _gsl_matrix_view w;
  w.matrix.size1 = n1; w.matrix.size2 = n2; w.matrix.tda = m->tda; w.matrix.data = m->data + i * m->tda + j;
  w.matrix.block = m->block; w.matrix.owner = 0;
  return w;
}
_gsl_matrix_const_view gsl_matrix_const_submatrix(const gsl_matrix *m, const size_t i, const size_t j, const size_t n1, const size_t n2) {
  if (i + n1 > m->size1 || j + n2 > m->size2) gsl_error("submatrix out of range", __FILE__, __LINE__, GSL_EINVAL);
  _gsl_matrix_const_view w;
  w.matrix.size1 = n1; w.matrix.size2 = n2; w.matrix.tda = m->tda; w.matrix.data = m->data + i * m->tda + j;
  w.matrix.block = m->block; w.matrix.owner = 0;
  return w;
}
_gsl_matrix_view gsl_matrix_view_array(double *base, const size_t n1, const size_t n2) {
  _gsl_matrix_view w;
  w.matrix.size1 = n1; w.matrix.size2 = n2; w.matrix.tda = n2; w.matrix.data = base; w.matrix.block = 0; w.matrix.owner = 0;
  return w;
}
_gsl_matrix_const_view gsl_matrix_const_view_array(const double *base, const size_t n1, const size_t n2) {
  _gsl_matrix_const_view w;
  w.matrix.size1 = n1; w.matrix.size2 = n2; w.matrix.tda = n2; w.matrix.data = (double *)base; w.matrix.block = 0; w.matrix.owner = 0;
  return w;
}

gsl_vector_int *gsl_vector_int_alloc(const size_t n) {
  gsl_vector_int *v = (gsl_vector_int *)malloc(sizeof(gsl_vector_int));
  gsl_block_int *b = (gsl_block_int *)malloc(sizeof(gsl_block_int));
  b->size = n; b->data = (int *)malloc((n ? n : 1) * sizeof(int));
  v->size = n; v->stride = 1; v->data = b->data; v->block = b; v->owner = 1;
  return v;
}
void gsl_vector_int_free(gsl_vector_int *v) { if (!v) return; if (v->owner) { free(v->block->data); free(v->block); } free(v); }
gsl_matrix_int *gsl_matrix_int_alloc(const size_t n1, const size_t n2) {
  gsl_matrix_int *m = (gsl_matrix_int *)malloc(sizeof(gsl_matrix_int));
  gsl_block_int *b = (gsl_block_int *)malloc(sizeof(gsl_block_int));
  b->size = n1 * n2; b->data = (int *)malloc((n1 * n2 ? n1 * n2 : 1) * sizeof(int));
  m->size1 = n1; m->size2 = n2; m->tda = n2; m->data = b->data; m->block = b; m->owner = 1;
  return m;
}
void gsl_matrix_int_free(gsl_matrix_int *m) { if (!m) return; if (m->owner) { free(m->block->data); free(m->block); } free(m); }

// ---------------------------------------------------------------- permutation
gsl_permutation *gsl_permutation_alloc(const size_t n) {
  gsl_permutation *p = (gsl_permutation *)malloc(sizeof(gsl_permutation));
  p->size = n; p->data = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
  return p;
}
void gsl_permutation_init(gsl_permutation *p) { for (size_t i = 0; i < p->size; i++) p->data[i] = i; }
gsl_permutation *gsl_permutation_calloc(const size_t n) { gsl_permutation *p = gsl_permutation_alloc(n); gsl_permutation_init(p); return p; }
void gsl_permutation_free(gsl_permutation *p) { if (!p) return; free(p->data); free(p); }

// ---------------------------------------------------------------- blas (row-major gsl_matrix -> CBLAS RowMajor)
int gsl_blas_ddot(const gsl_vector *X, const gsl_vector *Y, double *r) {
  if (X->size != Y->size) GSL_ERROR("invalid length", GSL_EBADLEN);
  *r = scipy_cblas_ddot((int)X->size, X->data, (int)X->stride, Y->data, (int)Y->stride);
  return GSL_SUCCESS;
}
int gsl_blas_daxpy(double a, const gsl_vector *X, gsl_vector *Y) {
  if (X->size != Y->size) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_daxpy((int)X->size, a, X->data, (int)X->stride, Y->data, (int)Y->stride);
  return GSL_SUCCESS;
}
int gsl_blas_dgemv(CBLAS_TRANSPOSE_t T, double alpha, const gsl_matrix *A, const gsl_vector *X, double beta, gsl_vector *Y) {
  const size_t M = A->size1, N = A->size2;
  if ((T == CblasNoTrans && N == X->size && M == Y->size) || (T == CblasTrans && M == X->size && N == Y->size)) {
    scipy_cblas_dgemv(CblasRowMajor, T, (int)M, (int)N, alpha, A->data, (int)A->tda, X->data, (int)X->stride, beta, Y->data, (int)Y->stride);
    return GSL_SUCCESS;
  }
  GSL_ERROR("invalid length", GSL_EBADLEN);
}
int gsl_blas_dger(double alpha, const gsl_vector *X, const gsl_vector *Y, gsl_matrix *A) {
  if (X->size != A->size1 || Y->size != A->size2) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_dger(CblasRowMajor, (int)A->size1, (int)A->size2, alpha, X->data, (int)X->stride, Y->data, (int)Y->stride, A->data, (int)A->tda);
  return GSL_SUCCESS;
}
int gsl_blas_dsyr(CBLAS_UPLO_t U, double alpha, const gsl_vector *X, gsl_matrix *A) {
  if (A->size1 != A->size2) GSL_ERROR("matrix must be square", GSL_ENOTSQR);
  if (X->size != A->size1) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_dsyr(CblasRowMajor, U, (int)A->size1, alpha, X->data, (int)X->stride, A->data, (int)A->tda);
  return GSL_SUCCESS;
}
int gsl_blas_dsyr2(CBLAS_UPLO_t U, double alpha, const gsl_vector *X, const gsl_vector *Y, gsl_matrix *A) {
  if (A->size1 != A->size2) GSL_ERROR("matrix must be square", GSL_ENOTSQR);
  if (X->size != A->size1 || Y->size != A->size1) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_dsyr2(CblasRowMajor, U, (int)A->size1, alpha, X->data, (int)X->stride, Y->data, (int)Y->stride, A->data, (int)A->tda);
  return GSL_SUCCESS;
}
int gsl_blas_dsyrk(CBLAS_UPLO_t U, CBLAS_TRANSPOSE_t T, double alpha, const gsl_matrix *A, double beta, gsl_matrix *C) {
  const size_t J = (T == CblasNoTrans) ? A->size1 : A->size2, K = (T == CblasNoTrans) ? A->size2 : A->size1;
  if (C->size1 != C->size2) GSL_ERROR("matrix C must be square", GSL_ENOTSQR);
  if (C->size1 != J) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_dsyrk(CblasRowMajor, U, T, (int)C->size1, (int)K, alpha, A->data, (int)A->tda, beta, C->data, (int)C->tda);
  return GSL_SUCCESS;
}
int gsl_blas_dgemm(CBLAS_TRANSPOSE_t TA, CBLAS_TRANSPOSE_t TB, double alpha, const gsl_matrix *A, const gsl_matrix *B, double beta, gsl_matrix *C) {
  const size_t M = C->size1, N = C->size2;
  const size_t MA = (TA == CblasNoTrans) ? A->size1 : A->size2, NA = (TA == CblasNoTrans) ? A->size2 : A->size1;
  const size_t MB = (TB == CblasNoTrans) ? B->size1 : B->size2, NB = (TB == CblasNoTrans) ? B->size2 : B->size1;
  if (M == MA && N == NB && NA == MB) {
    scipy_cblas_dgemm(CblasRowMajor, TA, TB, (int)M, (int)N, (int)NA, alpha, A->data, (int)A->tda, B->data, (int)B->tda, beta, C->data, (int)C->tda);
    return GSL_SUCCESS;
  }
  GSL_ERROR("invalid length", GSL_EBADLEN);
}
int gsl_blas_dtrsv(CBLAS_UPLO_t U, CBLAS_TRANSPOSE_t T, CBLAS_DIAG_t D, const gsl_matrix *A, gsl_vector *X) {
  if (A->size1 != A->size2) GSL_ERROR("matrix must be square", GSL_ENOTSQR);
  if (A->size2 != X->size) GSL_ERROR("invalid length", GSL_EBADLEN);
  scipy_cblas_dtrsv(CblasRowMajor, U, T, D, (int)A->size1, A->data, (int)A->tda, X->data, (int)X->stride);
  return GSL_SUCCESS;
}

// ---------------------------------------------------------------- linalg
// LU with partial pivoting, row by row (GSL linalg/lu.c, the unblocked form: pivot = first largest |a_ij|, i >= j)
int gsl_linalg_LU_decomp(gsl_matrix *A, gsl_permutation *p, int *signum) {
  if (A->size1 != A->size2) GSL_ERROR("LU decomposition requires square matrix", GSL_ENOTSQR);
  if (p->size != A->size1) GSL_ERROR("permutation length must match matrix size", GSL_EBADLEN);
  const size_t N = A->size1;
  *signum = 1;
  gsl_permutation_init(p);
  for (size_t j = 0; j + 1 < N; j++) {
    double max = fabs(gsl_matrix_get(A, j, j));
    size_t ip = j;
    for (size_t i = j + 1; i < N; i++) {
      double aij = fabs(gsl_matrix_get(A, i, j));
      if (aij > max) { max = aij; ip = i; }
    }
    if (ip != j) {
      for (size_t k = 0; k < N; k++) {
        double t = gsl_matrix_get(A, j, k);
        gsl_matrix_set(A, j, k, gsl_matrix_get(A, ip, k));
        gsl_matrix_set(A, ip, k, t);
      }
      size_t t = p->data[j]; p->data[j] = p->data[ip]; p->data[ip] = t;
      *signum = -(*signum);
    }
    double ajj = gsl_matrix_get(A, j, j);
    if (ajj != 0.0) {
      for (size_t i = j + 1; i < N; i++) {
        double aij = gsl_matrix_get(A, i, j) / ajj;
        gsl_matrix_set(A, i, j, aij);
        for (size_t k = j + 1; k < N; k++) gsl_matrix_set(A, i, k, gsl_matrix_get(A, i, k) - aij * gsl_matrix_get(A, j, k));
      }
    }
  }
  return GSL_SUCCESS;
}
static int lu_singular(const gsl_matrix *LU) {
  for (size_t i = 0; i < LU->size1; i++) if (gsl_matrix_get(LU, i, i) == 0.0) return 1;
  return 0;
}
static void lu_svx(const gsl_matrix *LU, const gsl_permutation *p, gsl_vector *x) {
  const size_t N = LU->size1;
  double *t = (double *)malloc(N * sizeof(double));
  for (size_t i = 0; i < N; i++) t[i] = gsl_vector_get(x, p->data[i]);  // x <- P b
  for (size_t i = 0; i < N; i++) {                                       // L y = P b (unit lower)
    double s = t[i];
    for (size_t k = 0; k < i; k++) s -= gsl_matrix_get(LU, i, k) * t[k];
    t[i] = s;
  }
  for (size_t ii = N; ii-- > 0;) {                                       // U x = y
    double s = t[ii];
    for (size_t k = ii + 1; k < N; k++) s -= gsl_matrix_get(LU, ii, k) * t[k];
    t[ii] = s / gsl_matrix_get(LU, ii, ii);
  }
  for (size_t i = 0; i < N; i++) gsl_vector_set(x, i, t[i]);
  free(t);
}
int gsl_linalg_LU_solve(const gsl_matrix *LU, const gsl_permutation *p, const gsl_vector *b, gsl_vector *x) {
  if (LU->size1 != LU->size2) GSL_ERROR("LU matrix must be square", GSL_ENOTSQR);
  if (LU->size1 != b->size || LU->size1 != x->size || LU->size1 != p->size) GSL_ERROR("matrix size must match vector sizes", GSL_EBADLEN);
  if (lu_singular(LU)) GSL_ERROR("matrix is singular", GSL_EDOM);
  gsl_vector_memcpy(x, b);
  lu_svx(LU, p, x);
  return GSL_SUCCESS;
}
int gsl_linalg_LU_invert(const gsl_matrix *LU, const gsl_permutation *p, gsl_matrix *inv) {
  if (LU->size1 != LU->size2 || inv->size1 != LU->size1 || inv->size2 != LU->size1) GSL_ERROR("matrix sizes must match", GSL_EBADLEN);
  if (lu_singular(LU)) GSL_ERROR("matrix is singular", GSL_EDOM);
  gsl_matrix_set_identity(inv);
  for (size_t j = 0; j < LU->size1; j++) {
    _gsl_vector_view c = gsl_matrix_column(inv, j);
    lu_svx(LU, p, &c.vector);
  }
  return GSL_SUCCESS;
}
double gsl_linalg_LU_det(gsl_matrix *LU, int signum) {
  double det = (double)signum;
  for (size_t i = 0; i < LU->size1; i++) det *= gsl_matrix_get(LU, i, i);
  return det;
}
double gsl_linalg_LU_lndet(gsl_matrix *LU) {
  double lndet = 0.0;
  for (size_t i = 0; i < LU->size1; i++) lndet += log(fabs(gsl_matrix_get(LU, i, i)));
  return lndet;
}
int gsl_linalg_cholesky_decomp1(gsl_matrix *A) {
  if (A->size1 != A->size2) GSL_ERROR("Cholesky decomposition requires square matrix", GSL_ENOTSQR);
  const size_t N = A->size1;
  for (size_t j = 0; j < N; j++) {
    double ajj = gsl_matrix_get(A, j, j);
    for (size_t k = 0; k < j; k++) ajj -= gsl_matrix_get(A, j, k) * gsl_matrix_get(A, j, k);
    if (!(ajj > 0.0)) GSL_ERROR("matrix is not positive definite", GSL_EDOM);
    ajj = sqrt(ajj);
    gsl_matrix_set(A, j, j, ajj);
    for (size_t i = j + 1; i < N; i++) {
      double s = gsl_matrix_get(A, i, j);
      for (size_t k = 0; k < j; k++) s -= gsl_matrix_get(A, i, k) * gsl_matrix_get(A, j, k);
      gsl_matrix_set(A, i, j, s / ajj);
    }
  }
  return GSL_SUCCESS;
}
int gsl_linalg_cholesky_decomp(gsl_matrix *A) {
  int s = gsl_linalg_cholesky_decomp1(A);
  if (s == GSL_SUCCESS)
    for (size_t i = 0; i < A->size1; i++) for (size_t j = i + 1; j < A->size2; j++) gsl_matrix_set(A, i, j, gsl_matrix_get(A, j, i));
  return s;
}
int gsl_linalg_QR_decomp(gsl_matrix *, gsl_vector *) { off_path("gsl_linalg_QR_decomp"); return GSL_EUNIMPL; }
int gsl_linalg_QR_solve(const gsl_matrix *, const gsl_vector *, const gsl_vector *, gsl_vector *) { off_path("gsl_linalg_QR_solve"); return GSL_EUNIMPL; }

// ---------------------------------------------------------------- eigen (values only; the reference uses it for K checks)
gsl_eigen_symm_workspace *gsl_eigen_symm_alloc(const size_t n) {
  gsl_eigen_symm_workspace *w = (gsl_eigen_symm_workspace *)malloc(sizeof(gsl_eigen_symm_workspace));
  w->size = n; w->d = 0; w->sd = 0;
  return w;
}
void gsl_eigen_symm_free(gsl_eigen_symm_workspace *w) { free(w); }
int gsl_eigen_symm(gsl_matrix *A, gsl_vector *eval, gsl_eigen_symm_workspace *) {
  if (A->size1 != A->size2) GSL_ERROR("matrix must be square to compute eigenvalues", GSL_ENOTSQR);
  if (eval->size != A->size1) GSL_ERROR("eigenvalue vector must match matrix size", GSL_EBADLEN);
  int n = (int)A->size1, lda = (int)A->tda, info = 0, lwork = -1;
  char jobz = 'N', uplo = 'U';  // row-major lower triangle == column-major upper triangle (GSL reads the lower one)
  double wq = 0;
  double *w = (double *)malloc(n * sizeof(double));
  scipy_dsyev_(&jobz, &uplo, &n, A->data, &lda, w, &wq, &lwork, &info);
  lwork = (int)wq;
  double *work = (double *)malloc((size_t)lwork * sizeof(double));
  scipy_dsyev_(&jobz, &uplo, &n, A->data, &lda, w, work, &lwork, &info);
  for (int i = 0; i < n; i++) gsl_vector_set(eval, i, w[i]);
  free(work); free(w);
  if (info != 0) GSL_ERROR("dsyev failed", GSL_EFAILED);
  return GSL_SUCCESS;
}

// ---------------------------------------------------------------- cdf
// GSL cdf/beta_inc.c: continued fraction (modified Lentz) for the incomplete beta function
static double beta_cont_frac(const double a, const double b, const double x, const double epsabs) {
  const unsigned int max_iter = 512;
  const double cutoff = 2.0 * GSL_DBL_MIN;
  unsigned int iter_count = 0;
  double cf;
  double num_term = 1.0;
  double den_term = 1.0 - (a + b) * x / (a + 1.0);
  if (fabs(den_term) < cutoff) den_term = GSL_NAN;
  den_term = 1.0 / den_term;
  cf = den_term;
  while (iter_count < max_iter) {
    const int k = iter_count + 1;
    double coeff = k * (b - k) * x / (((a - 1.0) + 2 * k) * (a + 2 * k));
    double delta_frac;
    den_term = 1.0 + coeff * den_term;
    num_term = 1.0 + coeff / num_term;
    if (fabs(den_term) < cutoff) den_term = GSL_NAN;
    if (fabs(num_term) < cutoff) num_term = GSL_NAN;
    den_term = 1.0 / den_term;
    delta_frac = den_term * num_term;
    cf *= delta_frac;
    coeff = -(a + k) * (a + b + k) * x / ((a + 2 * k) * (a + 2 * k + 1.0));
    den_term = 1.0 + coeff * den_term;
    num_term = 1.0 + coeff / num_term;
    if (fabs(den_term) < cutoff) den_term = GSL_NAN;
    if (fabs(num_term) < cutoff) num_term = GSL_NAN;
    den_term = 1.0 / den_term;
    delta_frac = den_term * num_term;
    cf *= delta_frac;
    if (fabs(delta_frac - 1.0) < 2.0 * GSL_DBL_EPSILON) break;
    if (cf * fabs(delta_frac - 1.0) < epsabs) break;
    ++iter_count;
  }
  if (iter_count >= max_iter) return GSL_NAN;
  return cf;
}
static double gamma_inc_Q(double a, double x);
static double gamma_inc_P(double a, double x);
// A * I_x(a,b) + Y
static double beta_inc_AXPY(const double A, const double Y, const double a, const double b, const double x) {
  if (x == 0.0) return A * 0 + Y;
  if (x == 1.0) return A * 1 + Y;
  if (a > 1e5 && b < 10 && x > a / (a + b)) {
    // Handle asymptotic regime, large a, small b, x > peak [AS 26.5.17]
    double N = a + (b - 1.0) / 2.0;
    return A * gamma_inc_Q(b, -N * log(x)) + Y;
  }
  if (b > 1e5 && a < 10 && x < b / (a + b)) {
    // Handle asymptotic regime, small a, large b, x < peak [AS 26.5.17]
    double N = b + (a - 1.0) / 2.0;
    return A * gamma_inc_P(a, -N * log1p(-x)) + Y;
  }
  double ln_beta = lgamma(a) + lgamma(b) - lgamma(a + b);
  double ln_pre = -ln_beta + a * log(x) + b * log1p(-x);
  double prefactor = exp(ln_pre);
  if (x < (a + 1.0) / (a + b + 2.0)) {
    double epsabs = fabs(Y / (A * prefactor / a)) * GSL_DBL_EPSILON;
    double cf = beta_cont_frac(a, b, x, epsabs);
    return A * (prefactor * cf / a) + Y;
  } else {
    double epsabs = fabs((A + Y) / (A * prefactor / b)) * GSL_DBL_EPSILON;
    double cf = beta_cont_frac(b, a, 1.0 - x, epsabs);
    double term = prefactor * cf / b;
    if (A == -Y) return -A * term;
    return A * (1 - term) + Y;
  }
}
double gsl_cdf_fdist_Q(const double x, const double nu1, const double nu2) {
  double r = nu2 / nu1;
  if (x < r) {
    double u = x / (r + x);
    return beta_inc_AXPY(-1.0, 1.0, nu1 / 2.0, nu2 / 2.0, u);
  }
  double u = r / (r + x);
  return beta_inc_AXPY(1.0, 0.0, nu2 / 2.0, nu1 / 2.0, u);
}
double gsl_cdf_fdist_P(const double x, const double nu1, const double nu2) {
  double r = nu2 / nu1;
  if (x < r) {
    double u = x / (r + x);
    return beta_inc_AXPY(1.0, 0.0, nu1 / 2.0, nu2 / 2.0, u);
  }
  double u = r / (r + x);
  return beta_inc_AXPY(-1.0, 1.0, nu2 / 2.0, nu1 / 2.0, u);
}
// regularised incomplete gamma functions: series for P, Legendre continued fraction (Lentz) for Q
static double gamma_inc_P_series(double a, double x) {
  double sum = 1.0 / a, term = 1.0 / a;
  for (int n = 1; n < 100000; n++) {
    term *= x / (a + n);
    sum += term;
    if (fabs(term) < fabs(sum) * 1e-17) break;
  }
  return sum * exp(-x + a * log(x) - lgamma(a));
}
static double gamma_inc_Q_cf(double a, double x) {
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; i++) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return exp(-x + a * log(x) - lgamma(a)) * h;
}
static double gamma_inc_Q(double a, double x) {
  if (x <= 0.0) return 1.0;
  if (a == 0.5) return erfc(sqrt(x));
  if (a == 1.0) return exp(-x);
  if (x < a + 1.0) return 1.0 - gamma_inc_P_series(a, x);
  return gamma_inc_Q_cf(a, x);
}
static double gamma_inc_P(double a, double x) {
  if (x <= 0.0) return 0.0;
  if (a == 0.5) return erf(sqrt(x));
  if (x < a + 1.0) return gamma_inc_P_series(a, x);
  return 1.0 - gamma_inc_Q_cf(a, x);
}
double gsl_cdf_chisq_Q(const double x, const double nu) { return (x <= 0.0) ? 1.0 : gamma_inc_Q(nu / 2.0, x / 2.0); }
double gsl_cdf_chisq_P(const double x, const double nu) { return (x <= 0.0) ? 0.0 : gamma_inc_P(nu / 2.0, x / 2.0); }
double gsl_cdf_chisq_Qinv(const double Q, const double nu) {
  if (Q >= 1.0) return 0.0;
  if (Q <= 0.0) return GSL_POSINF;
  double lo = 0.0, hi = GSL_MAX(1.0, nu);
  while (gsl_cdf_chisq_Q(hi, nu) > Q && hi < 1e300) hi *= 2.0;
  for (int it = 0; it < 400; it++) {
    double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (gsl_cdf_chisq_Q(mid, nu) > Q) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}
double gsl_cdf_gaussian_P(const double x, const double sigma) { return 0.5 * erfc(-x / (sigma * M_SQRT2)); }
double gsl_cdf_gaussian_Q(const double x, const double sigma) { return 0.5 * erfc(x / (sigma * M_SQRT2)); }
double gsl_sf_exp(const double x) { return exp(x); }
double gsl_sf_log_1plusx(const double x) { return log1p(x); }

// ---------------------------------------------------------------- roots (GSL roots/brent.c, newton.c, fsolver.c, fdfsolver.c, convergence.c)
typedef struct { double a, b, c, d, e; double fa, fb, fc; } brent_state_t;
#define SAFE_FUNC_CALL(f, x, yp) do { *yp = GSL_FN_EVAL(f, x); if (!gsl_finite(*yp)) GSL_ERROR("function value is not finite", GSL_EBADFUNC); } while (0)
static int brent_init(void *vstate, gsl_function *f, double *root, double x_lower, double x_upper) {
  brent_state_t *state = (brent_state_t *)vstate;
  double f_lower, f_upper;
  *root = 0.5 * (x_lower + x_upper);
  SAFE_FUNC_CALL(f, x_lower, &f_lower);
  SAFE_FUNC_CALL(f, x_upper, &f_upper);
  state->a = x_lower; state->fa = f_lower;
  state->b = x_upper; state->fb = f_upper;
  state->c = x_upper; state->fc = f_upper;
  state->d = x_upper - x_lower;
  state->e = x_upper - x_lower;
  if ((f_lower < 0.0 && f_upper < 0.0) || (f_lower > 0.0 && f_upper > 0.0)) GSL_ERROR("endpoints do not straddle y=0", GSL_EINVAL);
  return GSL_SUCCESS;
}
static int brent_iterate(void *vstate, gsl_function *f, double *root, double *x_lower, double *x_upper) {
  brent_state_t *state = (brent_state_t *)vstate;
  double tol, m;
  int ac_equal = 0;
  double a = state->a, b = state->b, c = state->c;
  double fa = state->fa, fb = state->fb, fc = state->fc;
  double d = state->d, e = state->e;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) { ac_equal = 1; c = a; fc = fa; d = b - a; e = b - a; }
  if (fabs(fc) < fabs(fb)) { ac_equal = 1; a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
  tol = 0.5 * GSL_DBL_EPSILON * fabs(b);
  m = 0.5 * (c - b);
  if (fb == 0) { *root = b; *x_lower = b; *x_upper = b; return GSL_SUCCESS; }
  if (fabs(m) <= tol) {
    *root = b;
    if (b < c) { *x_lower = b; *x_upper = c; } else { *x_lower = c; *x_upper = b; }
    return GSL_SUCCESS;
  }
  if (fabs(e) < tol || fabs(fa) <= fabs(fb)) {
    d = m; e = m;  // bisection
  } else {
    double p, q, r;
    double s = fb / fa;
    if (ac_equal) { p = 2 * m * s; q = 1 - s; }
    else { q = fa / fc; r = fb / fc; p = s * (2 * m * q * (q - r) - (b - a) * (r - 1)); q = (q - 1) * (r - 1) * (s - 1); }
    if (p > 0) q = -q; else p = -p;
    if (2 * p < GSL_MIN(3 * m * q - fabs(tol * q), fabs(e * q))) { e = d; d = p / q; }
    else { d = m; e = m; }
  }
  a = b; fa = fb;
  if (fabs(d) > tol) b += d; else b += (m > 0 ? +tol : -tol);
  SAFE_FUNC_CALL(f, b, &fb);
  state->a = a; state->b = b; state->c = c; state->d = d; state->e = e;
  state->fa = fa; state->fb = fb; state->fc = fc;
  *root = b;
  if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) c = a;
  if (b < c) { *x_lower = b; *x_upper = c; } else { *x_lower = c; *x_upper = b; }
  return GSL_SUCCESS;
}
static const gsl_root_fsolver_type brent_type = {"brent", sizeof(brent_state_t), &brent_init, &brent_iterate};
const gsl_root_fsolver_type *gsl_root_fsolver_brent = &brent_type;

gsl_root_fsolver *gsl_root_fsolver_alloc(const gsl_root_fsolver_type *T) {
  gsl_root_fsolver *s = (gsl_root_fsolver *)malloc(sizeof(gsl_root_fsolver));
  s->state = malloc(T->size);
  s->type = T; s->function = NULL; s->root = 0; s->x_lower = 0; s->x_upper = 0;
  return s;
}
void gsl_root_fsolver_free(gsl_root_fsolver *s) { if (!s) return; free(s->state); free(s); }
int gsl_root_fsolver_set(gsl_root_fsolver *s, gsl_function *f, double x_lower, double x_upper) {
  if (x_lower > x_upper) GSL_ERROR("invalid interval (lower > upper)", GSL_EINVAL);
  s->function = f; s->root = 0.5 * (x_lower + x_upper); s->x_lower = x_lower; s->x_upper = x_upper;
  return (s->type->set)(s->state, s->function, &(s->root), x_lower, x_upper);
}
int gsl_root_fsolver_iterate(gsl_root_fsolver *s) { return (s->type->iterate)(s->state, s->function, &(s->root), &(s->x_lower), &(s->x_upper)); }
const char *gsl_root_fsolver_name(const gsl_root_fsolver *s) { return s->type->name; }
double gsl_root_fsolver_root(const gsl_root_fsolver *s) { return s->root; }
double gsl_root_fsolver_x_lower(const gsl_root_fsolver *s) { return s->x_lower; }
double gsl_root_fsolver_x_upper(const gsl_root_fsolver *s) { return s->x_upper; }

typedef struct { double f, df; } newton_state_t;
static int newton_init(void *vstate, gsl_function_fdf *fdf, double *root) {
  newton_state_t *state = (newton_state_t *)vstate;
  const double x = *root;
  state->f = GSL_FN_FDF_EVAL_F(fdf, x);
  state->df = GSL_FN_FDF_EVAL_DF(fdf, x);
  return GSL_SUCCESS;
}
static int newton_iterate(void *vstate, gsl_function_fdf *fdf, double *root) {
  newton_state_t *state = (newton_state_t *)vstate;
  double root_new, f_new, df_new;
  if (state->df == 0.0) GSL_ERROR("derivative is zero", GSL_EZERODIV);
  root_new = *root - (state->f / state->df);
  *root = root_new;
  GSL_FN_FDF_EVAL_F_DF(fdf, root_new, &f_new, &df_new);
  state->f = f_new;
  state->df = df_new;
  if (!gsl_finite(f_new)) GSL_ERROR("function value is not finite", GSL_EBADFUNC);
  if (!gsl_finite(df_new)) GSL_ERROR("derivative value is not finite", GSL_EBADFUNC);
  return GSL_SUCCESS;
}
static const gsl_root_fdfsolver_type newton_type = {"newton", sizeof(newton_state_t), &newton_init, &newton_iterate};
const gsl_root_fdfsolver_type *gsl_root_fdfsolver_newton = &newton_type;
gsl_root_fdfsolver *gsl_root_fdfsolver_alloc(const gsl_root_fdfsolver_type *T) {
  gsl_root_fdfsolver *s = (gsl_root_fdfsolver *)malloc(sizeof(gsl_root_fdfsolver));
  s->state = malloc(T->size);
  s->type = T; s->fdf = NULL; s->root = 0;
  return s;
}
int gsl_root_fdfsolver_set(gsl_root_fdfsolver *s, gsl_function_fdf *f, double root) {
  s->fdf = f; s->root = root;
  return (s->type->set)(s->state, s->fdf, &(s->root));
}
int gsl_root_fdfsolver_iterate(gsl_root_fdfsolver *s) { return (s->type->iterate)(s->state, s->fdf, &(s->root)); }
void gsl_root_fdfsolver_free(gsl_root_fdfsolver *s) { if (!s) return; free(s->state); free(s); }
const char *gsl_root_fdfsolver_name(const gsl_root_fdfsolver *s) { return s->type->name; }
double gsl_root_fdfsolver_root(const gsl_root_fdfsolver *s) { return s->root; }
int gsl_root_test_interval(double x_lower, double x_upper, double epsabs, double epsrel) {
  const double abs_lower = fabs(x_lower), abs_upper = fabs(x_upper);
  double min_abs, tolerance;
  if (epsabs < 0.0) GSL_ERROR("absolute tolerance is negative", GSL_EBADTOL);
  if (epsrel < 0.0) GSL_ERROR("relative tolerance is negative", GSL_EBADTOL);
  if (x_lower > x_upper) GSL_ERROR("lower bound larger than upper bound", GSL_EINVAL);
  if ((x_lower > 0.0 && x_upper > 0.0) || (x_lower < 0.0 && x_upper < 0.0)) min_abs = GSL_MIN(abs_lower, abs_upper);
  else min_abs = 0;
  tolerance = epsabs + epsrel * min_abs;
  if (fabs(x_upper - x_lower) < tolerance) return GSL_SUCCESS;
  return GSL_CONTINUE;
}
int gsl_root_test_delta(double x1, double x0, double epsabs, double epsrel) {
  const double tolerance = epsabs + epsrel * fabs(x1);
  if (epsabs < 0.0) GSL_ERROR("absolute tolerance is negative", GSL_EBADTOL);
  if (epsrel < 0.0) GSL_ERROR("relative tolerance is negative", GSL_EBADTOL);
  if (fabs(x1 - x0) < tolerance || x1 == x0) return GSL_SUCCESS;
  return GSL_CONTINUE;
}

// ---------------------------------------------------------------- rng: MT19937 (GSL's default generator)
typedef struct { unsigned long mt[624]; int mti; } mt_state_t;
static void mt_set(void *vstate, unsigned long int s) {
  mt_state_t *st = (mt_state_t *)vstate;
  if (s == 0) s = 4357;
  st->mt[0] = s & 0xffffffffUL;
  for (int i = 1; i < 624; i++) {
    st->mt[i] = (1812433253UL * (st->mt[i - 1] ^ (st->mt[i - 1] >> 30)) + i);
    st->mt[i] &= 0xffffffffUL;
  }
  st->mti = 624;
}
static unsigned long mt_get(void *vstate) {
  mt_state_t *st = (mt_state_t *)vstate;
  unsigned long *mt = st->mt;
  const unsigned long UPPER = 0x80000000UL, LOWER = 0x7fffffffUL;
#define MAGIC(y) (((y) & 0x1) ? 0x9908b0dfUL : 0)
  if (st->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; kk++) { unsigned long y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ MAGIC(y); }
    for (; kk < 623; kk++) { unsigned long y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ MAGIC(y); }
    { unsigned long y = (mt[623] & UPPER) | (mt[0] & LOWER); mt[623] = mt[396] ^ (y >> 1) ^ MAGIC(y); }
    st->mti = 0;
  }
  unsigned long k = mt[st->mti];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680UL;
  k ^= (k << 15) & 0xefc60000UL;
  k ^= (k >> 18);
  st->mti++;
  return k & 0xffffffffUL;
}
static double mt_get_double(void *vstate) { return mt_get(vstate) / 4294967296.0; }
static const gsl_rng_type mt_type = {"mt19937", 0xffffffffUL, 0, sizeof(mt_state_t), &mt_set, &mt_get, &mt_get_double};
const gsl_rng_type *gsl_rng_mt19937 = &mt_type;
const gsl_rng_type *gsl_rng_default = &mt_type;
unsigned long int gsl_rng_default_seed = 0;
const gsl_rng_type *gsl_rng_env_setup(void) {
  const char *p = getenv("GSL_RNG_SEED");
  if (p) gsl_rng_default_seed = strtoul(p, 0, 0);
  return gsl_rng_default;
}
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T) {
  gsl_rng *r = (gsl_rng *)malloc(sizeof(gsl_rng));
  r->state = calloc(1, T->size);
  r->type = T;
  gsl_rng_set(r, gsl_rng_default_seed);
  return r;
}
void gsl_rng_free(gsl_rng *r) { if (!r) return; free(r->state); free(r); }
void gsl_rng_set(const gsl_rng *r, unsigned long int seed) { (r->type->set)(r->state, seed); }
const char *gsl_rng_name(const gsl_rng *r) { return r->type->name; }
unsigned long int gsl_rng_get(const gsl_rng *r) { return (r->type->get)(r->state); }
double gsl_rng_uniform(const gsl_rng *r) { return (r->type->get_double)(r->state); }
unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n) {
  unsigned long int offset = r->type->min, range = r->type->max - offset, scale, k;
  if (n > range || n == 0) GSL_ERROR_VAL("invalid n, either 0 or exceeds maximum value of generator", GSL_EINVAL, 0);
  scale = range / n;
  do { k = (((r->type->get)(r->state)) - offset) / scale; } while (k >= n);
  return k;
}
int gsl_ran_choose(const gsl_rng *r, void *dest, size_t k, void *src, size_t n, size_t size) {
  size_t i, j = 0;
  if (k > n) GSL_ERROR("k is greater than n, cannot sample more than n items", GSL_EINVAL);
  for (i = 0; i < n && j < k; i++) {
    if ((n - i) * gsl_rng_uniform(r) < k - j) { memcpy((char *)dest + size * j, (char *)src + size * i, size); j++; }
  }
  return GSL_SUCCESS;
}
gsl_ran_discrete_t *gsl_ran_discrete_preproc(size_t, const double *) { off_path("gsl_ran_discrete_preproc"); return 0; }
size_t gsl_ran_discrete(const gsl_rng *, const gsl_ran_discrete_t *) { off_path("gsl_ran_discrete"); return 0; }
void gsl_ran_discrete_free(gsl_ran_discrete_t *) {}
double gsl_ran_gamma(const gsl_rng *, const double, const double) { off_path("gsl_ran_gamma"); return 0; }
double gsl_ran_gaussian(const gsl_rng *, const double) { off_path("gsl_ran_gaussian"); return 0; }
double gsl_ran_geometric_pdf(const unsigned int, const double) { off_path("gsl_ran_geometric_pdf"); return 0; }

// ---------------------------------------------------------------- multiroots (VC only)
static const gsl_multiroot_fdfsolver_type hybridsj_type = {"hybridsj"};
const gsl_multiroot_fdfsolver_type *gsl_multiroot_fdfsolver_hybridsj = &hybridsj_type;
gsl_multiroot_fdfsolver *gsl_multiroot_fdfsolver_alloc(const gsl_multiroot_fdfsolver_type *, size_t) { off_path("gsl_multiroot_fdfsolver"); return 0; }
void gsl_multiroot_fdfsolver_free(gsl_multiroot_fdfsolver *) {}
int gsl_multiroot_fdfsolver_set(gsl_multiroot_fdfsolver *, gsl_multiroot_function_fdf *, const gsl_vector *) { off_path("gsl_multiroot_fdfsolver_set"); return 0; }
int gsl_multiroot_fdfsolver_iterate(gsl_multiroot_fdfsolver *) { off_path("gsl_multiroot_fdfsolver_iterate"); return 0; }
int gsl_multiroot_test_residual(const gsl_vector *, double) { off_path("gsl_multiroot_test_residual"); return 0; }

}  // extern "C"
