// TEST DOUBLE of the C ABI (include/gemma_hip.h) over the CPU oracle -- builds into a library with the product's
// soname so that the C++ host layer (gemma_host.hpp, gemma_io_host.hpp, tests/cpp/gemma_file_driver.cpp) can be run
// end to end in a container without a GPU ("a CPU implementation of the same ABI (the oracle) is the test
// double").  It lives under tests/, is built only by tests/test_file_driver_cpu.py into a temporary directory, and is
// never installed next to the product: gemma_amd/ has no CPU path and keeps failing with GEMMA_HIP_ENODEV without a
// device.  Only the entry points the file driver reaches are implemented; everything else is absent on purpose
// (an unexpected call is a link error, not a silent fallback).
//
// Arithmetic: oracle/gemma_oracle.c (orc_*: restatement of the reference, pinned on the reference binary's outputs)
// for imputation, centring, kinship preparation, the null model and the per-SNP statistics; plain loops for the
// GEMMs; for the symmetric eigenproblem LAPACK's dsyev from the OpenBLAS inside scipy when the test names it in
// GEMMA_DOUBLE_LAPACK (dlopen), otherwise a cyclic Jacobi sweep (the statistics do not depend on the eigenbasis,
// SURVEY App. A.6); the first-pass SNP filters restated here from src/gemma_io.cpp:753-853 / :942-1049, the exact HWE
// test from src/mathfunc.cpp:546-640 (Wigginton et al. 2005).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "gemma_hip.h"
#include "../../gemma_amd/csrc/comm_shm.hpp" // the host shared-memory transport (pure POSIX) the library uses for its tests too

extern "C" {
typedef struct {
  double beta, se, lambda_remle, lambda_mle, p_wald, p_lrt, p_score, logl_H1;
} orc_sumstat;
void orc_lmm_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtW, const double *Uty,
                   const double *UtX, size_t l, double l_min, double l_max, size_t n_region, double l_mle_null,
                   double logl_mle_H0, int plink_nan_rule, double *carry, orc_sumstat *out, long *diag);
void orc_gene_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtW, const double *Utx,
                    const double *UtY, size_t l, double l_min, double l_max, size_t n_region, orc_sumstat *out);
void orc_gxe_batch(int a_mode, size_t n, size_t c, const double *eval, const double *UtWe, const double *Uty,
                   const double *UtX, const double *UtZ, const int *flip, size_t l, double l_min, double l_max,
                   size_t n_region, double l_mle_null, orc_sumstat *out);
void orc_lm_batch(int a_mode, size_t n, size_t c, const double *W, const double *WtWi, const double *y, const double *X,
                  size_t l, orc_sumstat *out);
void orc_impute_mean(double *X, size_t l, size_t n);
void orc_kin_prepare(double *X, size_t l, size_t n, int k_mode);
size_t orc_bed_decode(const unsigned char *bytes, size_t ni_total, const int *indicator, double *x);
void orc_CenterMatrix(double *G, size_t n);
double orc_zero_small_eval(double *eval, size_t n);
void orc_CalcLambda_null(char func_name, size_t n, size_t c, const double *eval, const double *UtW, const double *Uty,
                         double l_min, double l_max, size_t n_region, double *lambda, double *logl_H0);
void orc_CalcPve(size_t n, size_t c, const double *eval, const double *UtW, const double *Uty, double lambda,
                 double trace_G, double *pve, double *pve_se);
typedef struct {
  size_t em_iter, nr_iter, n_region;
  double em_prec, nr_prec, l_min, l_max, p_nr;
  size_t crt;
} orc_mv_cfg;
void orc_mvlmm_null(const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval, const double *W,
                    const double *Y, double *Vg_remle, double *Ve_remle, double *B_remle, double *logl_remle,
                    double *Vg_mle, double *Ve_mle, double *B_mle, double *logl_mle);
void orc_mvlmm_batch(int a_mode, const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval, const double *W,
                     const double *Y, const double *UtX, size_t l, const double *Vg_null, const double *Ve_null,
                     const double *B_null, double logl_H0, double *out);
void orc_mvlmm_batch_gxe(int a_mode, const orc_mv_cfg *cfg, size_t n, size_t d, size_t cw, const double *eval, const double *W,
                         const double *Y, const double *UtX, const double *UtX2, size_t l, const double *Vg_null,
                         const double *Ve_null, const double *B_null, double *out);
void orc_CalcLmmVgVeBeta(size_t n, size_t c, const double *eval, const double *UtW, const double *Uty, double lambda,
                         double *vg, double *ve, double *beta, double *se_beta);
}

namespace {
std::string g_err;
int fail(int code, const char *msg) {
  g_err = msg;
  return code;
}
const double NaN = std::numeric_limits<double>::quiet_NaN();

struct Kin {
  size_t n = 0, ns = 0;
  int k_mode = 1;
  std::vector<double> K;
} g_kin;

struct Lmm {
  bool on = false;
  gemma_lmm_cfg cfg;
  std::vector<double> U, eval, UtW, Uty, carry;
  std::vector<int> ind;
  std::vector<double> env, UtWe; // -gxe: environment variable, [U^T W | U^T env] (n x (c + 1))
  // multivariate: phenotypes and covariates transposed (d x n, c x n: the oracle's layout), null fit, options
  size_t d = 0;
  std::vector<double> Yt, Wt;
  gemma_mvlmm_null mv_null;
  orc_mv_cfg mv_cfg;
  bool mv_gxe = false;
} g_lmm;

struct Lm {
  bool on = false;
  int a_mode = 51;
  size_t n = 0, c = 0;
  std::vector<double> W, WtWi, y;
} g_lm;

std::vector<double> transposed(const double *A, size_t rows, size_t cols) {
  std::vector<double> T(rows * cols);
  for (size_t i = 0; i < rows; ++i)
    for (size_t j = 0; j < cols; ++j) T[j * rows + i] = A[i * cols + j];
  return T;
}

// exact test of Hardy-Weinberg equilibrium: probability of every heterozygote count given the allele counts, built
// outwards from the most likely count; p = total probability of the counts no likelier than the observed one
double calc_hwe(size_t n_hom1, size_t n_hom2, size_t n_ab) {
  if (n_hom1 + n_hom2 + n_ab == 0) return 1.0;
  const long n_aa = (long)std::min(n_hom1, n_hom2), n_bb = (long)std::max(n_hom1, n_hom2), nab = (long)n_ab;
  const long rare = 2 * n_aa + nab, genotypes = nab + n_bb + n_aa;
  std::vector<double> het((size_t)rare + 1, 0.0);
  long mid = rare * (2 * genotypes - rare) / (2 * genotypes);
  if ((rare & 1) ^ (mid & 1)) mid++;
  het[(size_t)mid] = 1.0;
  double sum = 1.0;
  long homr = (rare - mid) / 2, homc = genotypes - mid - homr;
  for (long h = mid; h > 1; h -= 2) {
    het[(size_t)h - 2] = het[(size_t)h] * (double)h * ((double)h - 1.0) / (4.0 * ((double)homr + 1.0) * ((double)homc + 1.0));
    sum += het[(size_t)h - 2];
    homr++;
    homc++;
  }
  homr = (rare - mid) / 2;
  homc = genotypes - mid - homr;
  for (long h = mid; h <= rare - 2; h += 2) {
    het[(size_t)h + 2] = het[(size_t)h] * 4.0 * (double)homr * (double)homc / (((double)h + 2.0) * ((double)h + 1.0));
    sum += het[(size_t)h + 2];
    homr--;
    homc--;
  }
  double p = 0.0;
  for (long i = 0; i <= rare; ++i) {
    const double v = het[(size_t)i] / sum;
    if (v > het[(size_t)nab] / sum) continue;
    p += v;
  }
  return p > 1.0 ? 1.0 : p;
}

// rows of l SNPs over `n_out` individuals (NaN = missing) from either encoding
void decode(int kind, const void *geno, size_t l, size_t ld, const int *ind, size_t ni_total, size_t n_out,
            std::vector<double> &X) {
  X.assign(l * n_out, 0.0);
  for (size_t s = 0; s < l; ++s) {
    double *x = &X[s * n_out];
    if (kind == GEMMA_GENO_PLINK_2BIT) {
      orc_bed_decode(static_cast<const unsigned char *>(geno) + s * ld, ni_total, ind, x);
    } else {
      const double *g = static_cast<const double *>(geno) + s * ld;
      size_t o = 0;
      for (size_t i = 0; i < ni_total; ++i)
        if (!ind || ind[i]) x[o++] = g[i];
    }
  }
}
} // namespace

extern "C" {

int gemma_hip_init(int, int) { return GEMMA_HIP_OK; }
void gemma_hip_shutdown(void) {}
int gemma_hip_abi_version(void) { return GEMMA_HIP_ABI_VERSION; }
const char *gemma_hip_strerror(int code) { return code == 0 ? "ok" : "error (test double)"; }
const char *gemma_hip_last_error(void) { return g_err.c_str(); }
int gemma_hip_device_info(char *name, size_t len, int *n_cu, size_t *hbm_bytes) {
  if (name && len) snprintf(name, len, "CPU test double over the oracle");
  if (n_cu) *n_cu = 0;
  if (hbm_bytes) *hbm_bytes = 0;
  return GEMMA_HIP_OK;
}

int gemma_hip_dgemm(char ta, char tb, size_t M, size_t N, size_t K, double alpha, const double *A, size_t lda,
                    const double *B, size_t ldb, double beta, double *C, size_t ldc) {
  const bool tA = ta == 'T' || ta == 't', tB = tb == 'T' || tb == 't';
  std::vector<double> acc(N);
  for (size_t i = 0; i < M; ++i) {
    std::fill(acc.begin(), acc.end(), 0.0);
    for (size_t k = 0; k < K; ++k) {
      const double a = tA ? A[k * lda + i] : A[i * lda + k];
      for (size_t j = 0; j < N; ++j) acc[j] += a * (tB ? B[j * ldb + k] : B[k * ldb + j]);
    }
    for (size_t j = 0; j < N; ++j) C[i * ldc + j] = alpha * acc[j] + (beta == 0.0 ? 0.0 : beta * C[i * ldc + j]);
  }
  return GEMMA_HIP_OK;
}

int gemma_hip_kin_begin(size_t n_total, int k_mode) {
  g_kin.n = n_total;
  g_kin.ns = 0;
  g_kin.k_mode = k_mode;
  g_kin.K.assign(n_total * n_total, 0.0);
  return GEMMA_HIP_OK;
}
int gemma_hip_kin_add(int kind, const void *geno, size_t l, size_t ld) {
  if (kind != GEMMA_GENO_F64_SNP_MAJOR && kind != GEMMA_GENO_PLINK_2BIT) return fail(GEMMA_HIP_EINVAL, "kin_add kind");
  const size_t n = g_kin.n;
  std::vector<double> X;
  decode(kind, geno, l, ld, nullptr, n, n, X);
  orc_kin_prepare(X.data(), l, n, g_kin.k_mode);
  for (size_t s = 0; s < l; ++s) {
    const double *x = &X[s * n];
    for (size_t i = 0; i < n; ++i) {
      const double xi = x[i];
      double *k = &g_kin.K[i * n];
      for (size_t j = 0; j < n; ++j) k[j] += xi * x[j];
    }
  }
  g_kin.ns += l;
  return GEMMA_HIP_OK;
}
int gemma_hip_kin_end(double *K, size_t *ns_used) {
  for (size_t i = 0; i < g_kin.n * g_kin.n; ++i) K[i] = g_kin.K[i] / (double)g_kin.ns;
  if (ns_used) *ns_used = g_kin.ns;
  return GEMMA_HIP_OK;
}

// ReadFile_geno src/gemma_io.cpp:753-853 / ReadFile_bed :942-1049, statistics over the analysed individuals
int gemma_hip_snp_qc(int kind, const void *geno, size_t l, size_t ld, const int *indicator_idv, size_t ni_total,
                     const double *W, size_t n, size_t c, const gemma_qc_cfg *cfg, int *indicator_snp, double *maf_out,
                     size_t *n_miss_out) {
  std::vector<double> X;
  decode(kind, geno, l, ld, indicator_idv, ni_total, n, X);
  // (W^T W)^-1 by Gauss-Jordan
  std::vector<double> A(c * 2 * c, 0.0);
  for (size_t a = 0; a < c; ++a) {
    for (size_t b = 0; b < c; ++b)
      for (size_t i = 0; i < n; ++i) A[a * 2 * c + b] += W[i * c + a] * W[i * c + b];
    A[a * 2 * c + c + a] = 1.0;
  }
  for (size_t p = 0; p < c; ++p) {
    size_t piv = p;
    for (size_t r = p + 1; r < c; ++r)
      if (std::fabs(A[r * 2 * c + p]) > std::fabs(A[piv * 2 * c + p])) piv = r;
    for (size_t j = 0; j < 2 * c; ++j) std::swap(A[p * 2 * c + j], A[piv * 2 * c + j]);
    const double d = A[p * 2 * c + p];
    for (size_t j = 0; j < 2 * c; ++j) A[p * 2 * c + j] /= d;
    for (size_t r = 0; r < c; ++r) {
      if (r == p) continue;
      const double f = A[r * 2 * c + p];
      for (size_t j = 0; j < 2 * c; ++j) A[r * 2 * c + j] -= f * A[p * 2 * c + j];
    }
  }
  for (size_t s = 0; s < l; ++s) {
    double *x = &X[s * n];
    double sum = 0.0, first = NaN;
    size_t n_miss = 0, n0 = 0, n1 = 0, n2 = 0;
    bool differ = false;
    for (size_t i = 0; i < n; ++i) {
      const double g = x[i];
      if (g != g) { ++n_miss; continue; }
      if (g >= 0 && g <= 0.5) ++n0;
      if (g > 0.5 && g < 1.5) ++n1;
      if (g >= 1.5 && g <= 2.0) ++n2;
      if (first != first) first = g; else if (g != first) differ = true;
      sum += g;
    }
    const double maf = sum / (2.0 * (double)(n - n_miss));
    if (maf_out) maf_out[s] = maf;
    if (n_miss_out) n_miss_out[s] = n_miss;
    int keep = 1;
    if ((double)n_miss / (double)n > cfg->miss_level) keep = 0;
    else if ((maf < cfg->maf_level || maf > 1.0 - cfg->maf_level) && cfg->maf_level != -1) keep = 0;
    else if (kind == GEMMA_GENO_PLINK_2BIT ? ((n0 + n1) == 0 || (n1 + n2) == 0 || (n2 + n0) == 0) : !differ) keep = 0;
    else if (cfg->hwe_level != 0 && cfg->maf_level != -1 && calc_hwe(n0, n2, n1) < cfg->hwe_level) keep = 0;
    else if (c != 1) {
      std::vector<double> Wtx(c, 0.0);
      double v_x = 0.0, v_w = 0.0;
      for (size_t i = 0; i < n; ++i) {
        const double g = x[i] != x[i] ? maf * 2.0 : x[i];
        v_x += g * g;
        for (size_t a = 0; a < c; ++a) Wtx[a] += W[i * c + a] * g;
      }
      for (size_t a = 0; a < c; ++a) {
        double t = 0.0;
        for (size_t b = 0; b < c; ++b) t += A[a * 2 * c + c + b] * Wtx[b];
        v_w += Wtx[a] * t;
      }
      if (v_w / v_x > cfg->r2_level) keep = 0;
    }
    indicator_snp[s] = keep;
  }
  return GEMMA_HIP_OK;
}

int gemma_hip_center(double *G, size_t n) {
  orc_CenterMatrix(G, n);
  return GEMMA_HIP_OK;
}

// eigenvalues ascending, eigenvector k = column k of row-major U; EigenDecomp_Zeroed's clean-up
int gemma_hip_eigh(double *G, size_t n, double *U, double *eval, double *trace_G) {
  typedef void (*dsyev_t)(char *, char *, int *, double *, int *, double *, double *, int *, int *);
  static dsyev_t dsyev = nullptr;
  static bool looked = false;
  if (!looked) {
    looked = true;
    const char *path = getenv("GEMMA_DOUBLE_LAPACK");
    void *h = path ? dlopen(path, RTLD_NOW | RTLD_GLOBAL) : nullptr;
    if (h) dsyev = reinterpret_cast<dsyev_t>(dlsym(h, "scipy_dsyev_"));
  }
  if (dsyev) {
    char jobz = 'V', uplo = 'L';
    int N = (int)n, lwork = -1, info = 0;
    double wq = 0;
    dsyev(&jobz, &uplo, &N, G, &N, eval, &wq, &lwork, &info);
    lwork = (int)wq;
    std::vector<double> work((size_t)lwork);
    dsyev(&jobz, &uplo, &N, G, &N, eval, work.data(), &lwork, &info);
    if (info != 0) return fail(GEMMA_HIP_ENOCONV, "dsyev");
    for (size_t k = 0; k < n; ++k) // column-major eigenvector k = G[k * n + i]
      for (size_t i = 0; i < n; ++i) U[i * n + k] = G[k * n + i];
    *trace_G = orc_zero_small_eval(eval, n);
    return GEMMA_HIP_OK;
  }
  std::vector<double> V(n * n, 0.0);
  for (size_t i = 0; i < n; ++i) V[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (size_t p = 0; p < n; ++p)
      for (size_t q = p + 1; q < n; ++q) off += G[p * n + q] * G[p * n + q];
    if (off < 1e-30) break;
    for (size_t p = 0; p < n; ++p)
      for (size_t q = p + 1; q < n; ++q) {
        const double apq = G[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (G[q * n + q] - G[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (size_t k = 0; k < n; ++k) {
          const double gkp = G[k * n + p], gkq = G[k * n + q];
          G[k * n + p] = cs * gkp - sn * gkq;
          G[k * n + q] = sn * gkp + cs * gkq;
        }
        for (size_t k = 0; k < n; ++k) {
          const double gpk = G[p * n + k], gqk = G[q * n + k];
          G[p * n + k] = cs * gpk - sn * gqk;
          G[q * n + k] = sn * gpk + cs * gqk;
        }
        for (size_t k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = cs * vkp - sn * vkq;
          V[k * n + q] = sn * vkp + cs * vkq;
        }
      }
  }
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return G[a * n + a] < G[b * n + b]; });
  for (size_t k = 0; k < n; ++k) {
    eval[k] = G[order[k] * n + order[k]];
    for (size_t i = 0; i < n; ++i) U[i * n + k] = V[i * n + order[k]];
  }
  *trace_G = orc_zero_small_eval(eval, n);
  return GEMMA_HIP_OK;
}

int gemma_hip_lmm_null(size_t n, size_t c, const double *eval, const double *UtW, const double *Uty, double l_min,
                       double l_max, size_t n_region, double trace_G, double *o) {
  orc_CalcLambda_null('L', n, c, eval, UtW, Uty, l_min, l_max, n_region, &o[0], &o[1]);
  orc_CalcLambda_null('R', n, c, eval, UtW, Uty, l_min, l_max, n_region, &o[2], &o[3]);
  orc_CalcPve(n, c, eval, UtW, Uty, o[2], trace_G, &o[4], &o[5]);
  std::vector<double> beta(c), se(c);
  orc_CalcLmmVgVeBeta(n, c, eval, UtW, Uty, o[2], &o[6], &o[7], beta.data(), se.data());
  return GEMMA_HIP_OK;
}

int gemma_hip_lmm_setup(const gemma_lmm_cfg *cfg, const double *U, const double *eval, const double *UtW,
                        const double *Uty) {
  const size_t n = cfg->n, c = cfg->n_cvt;
  g_lmm.on = true;
  g_lmm.cfg = *cfg;
  g_lmm.U.assign(U, U + n * n);
  g_lmm.eval.assign(eval, eval + n);
  g_lmm.UtW.assign(UtW, UtW + n * c);
  g_lmm.Uty.assign(Uty, Uty + n);
  g_lmm.carry.assign(2, 0.0);
  g_lmm.ind.clear();
  g_lmm.env.clear();
  g_lmm.d = 0;
  return GEMMA_HIP_OK;
}
int gemma_hip_lmm_set_indicator(const int *ind, size_t ni_total) {
  if (!g_lmm.on && !g_lm.on) return fail(GEMMA_HIP_ESTATE, "lmm_set_indicator before lmm_setup / lm_setup");
  g_lmm.ind.assign(ind, ind + (ind ? ni_total : 0));
  return GEMMA_HIP_OK;
}
int gemma_hip_lmm_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  if (!g_lmm.on) return fail(GEMMA_HIP_ESTATE, "lmm_batch before lmm_setup");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  std::vector<double> X;
  if (kind == GEMMA_GENO_PLINK_2BIT)
    decode(kind, geno, l, ld, g_lmm.ind.empty() ? nullptr : g_lmm.ind.data(), g_lmm.ind.empty() ? n : g_lmm.ind.size(), n, X);
  else if (kind == GEMMA_GENO_F64_SNP_MAJOR)
    decode(kind, geno, l, ld, nullptr, n, n, X);
  else
    return fail(GEMMA_HIP_EINVAL, "lmm_batch kind");
  orc_impute_mean(X.data(), l, n);
  std::vector<double> UtX(l * n);
  gemma_hip_dgemm('N', 'N', l, n, n, 1.0, X.data(), n, g_lmm.U.data(), n, 0.0, UtX.data(), n); // row s = (U^T x_s)^T
  static_assert(sizeof(orc_sumstat) == sizeof(gemma_sumstat), "SUMSTAT layout");
  orc_lmm_batch(g_lmm.cfg.a_mode, n, c, g_lmm.eval.data(), g_lmm.UtW.data(), g_lmm.Uty.data(), UtX.data(), l,
                g_lmm.cfg.l_min, g_lmm.cfg.l_max, g_lmm.cfg.n_region, g_lmm.cfg.l_mle_null, g_lmm.cfg.logl_mle_H0,
                g_lmm.cfg.plink_nan_rule, g_lmm.carry.data(), reinterpret_cast<orc_sumstat *>(out), nullptr);
  return GEMMA_HIP_OK;
}
// pipelined form: the double computes at submit and hands the records over at collect (same ordering rules)
namespace {
std::vector<std::vector<gemma_sumstat>> g_pipe;
}
int gemma_hip_lmm_batch_submit(int kind, const void *geno, size_t l, size_t ld) {
  if (g_pipe.size() >= 2) return fail(GEMMA_HIP_ESTATE, "lmm_batch_submit: two blocks already in flight");
  std::vector<gemma_sumstat> out(l);
  const int rc = gemma_hip_lmm_batch(kind, geno, l, ld, out.data());
  if (rc) return rc;
  g_pipe.push_back(out);
  return GEMMA_HIP_OK;
}
int gemma_hip_lmm_batch_collect(gemma_sumstat *out, size_t *l) {
  if (g_pipe.empty()) return fail(GEMMA_HIP_ESTATE, "lmm_batch_collect: nothing in flight");
  std::copy(g_pipe.front().begin(), g_pipe.front().end(), out);
  if (l) *l = g_pipe.front().size();
  g_pipe.erase(g_pipe.begin());
  return GEMMA_HIP_OK;
}
int gemma_hip_lmm_finish(double *t_utx, double *t_opt) {
  g_pipe.clear();
  g_lmm.on = false;
  g_lmm.mv_gxe = false;
  if (t_utx) *t_utx = 0.0;
  if (t_opt) *t_opt = 0.0;
  return GEMMA_HIP_OK;
}

// ---- multivariate LMM over oracle/mvlmm_oracle.c ----------------------------------------------------------------
int gemma_hip_mvlmm_null(size_t n, size_t c, size_t d, const double *eval, const double *UtW, const double *UtY, double l_min,
                         double l_max, size_t n_region, const gemma_mvlmm_opt *opt, gemma_mvlmm_null *out) {
  const orc_mv_cfg cfg = {opt->em_iter, opt->nr_iter, n_region, opt->em_prec, opt->nr_prec, l_min, l_max, opt->p_nr, opt->crt};
  const std::vector<double> Wt = transposed(UtW, n, c), Yt = transposed(UtY, n, d);
  memset(out, 0, sizeof(*out));
  orc_mvlmm_null(&cfg, n, d, c, eval, Wt.data(), Yt.data(), out->Vg_remle, out->Ve_remle, out->B_remle, &out->logl_remle_H0,
                 out->Vg_mle, out->Ve_mle, out->B_mle, &out->logl_mle_H0);
  return GEMMA_HIP_OK;
}
int gemma_hip_mvlmm_set(size_t d, const double *UtY, const gemma_mvlmm_null *null_fit, const gemma_mvlmm_opt *opt) {
  if (!g_lmm.on) return fail(GEMMA_HIP_ESTATE, "mvlmm_set before lmm_setup");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  g_lmm.d = d;
  g_lmm.Yt = transposed(UtY, n, d);
  g_lmm.Wt = transposed(g_lmm.UtW.data(), n, c);
  g_lmm.mv_null = *null_fit;
  const orc_mv_cfg cfg = {opt->em_iter, opt->nr_iter, g_lmm.cfg.n_region, opt->em_prec, opt->nr_prec, g_lmm.cfg.l_min,
                          g_lmm.cfg.l_max, opt->p_nr, opt->crt};
  g_lmm.mv_cfg = cfg;
  g_lmm.mv_gxe = opt->gxe == 1;
  if (g_lmm.mv_gxe) {
    if (g_lmm.env.empty()) return fail(GEMMA_HIP_ESTATE, "mvlmm_set with gxe before lmm_set_env");
    g_lmm.Wt = transposed(g_lmm.UtWe.data(), n, c + 1); // (W, env)
  }
  return GEMMA_HIP_OK;
}
int gemma_hip_mvlmm_batch(int kind, const void *geno, size_t l, size_t ld, double *out) {
  if (!g_lmm.on || g_lmm.d == 0) return fail(GEMMA_HIP_ESTATE, "mvlmm_batch before mvlmm_set");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  std::vector<double> X;
  if (kind == GEMMA_GENO_PLINK_2BIT)
    decode(kind, geno, l, ld, g_lmm.ind.empty() ? nullptr : g_lmm.ind.data(), g_lmm.ind.empty() ? n : g_lmm.ind.size(), n, X);
  else
    decode(kind, geno, l, ld, nullptr, n, n, X);
  if (g_lmm.mv_gxe) { // MVLMM::AnalyzePlinkGXE, src/mvlmm.cpp:4416-4870
    std::vector<int> flip(l, 0);
    for (size_t s = 0; s < l; ++s) {
      double tot = 0.0;
      size_t cnt = 0;
      for (size_t i = 0; i < n; ++i)
        if (X[s * n + i] == X[s * n + i]) { tot += X[s * n + i]; ++cnt; }
      flip[s] = cnt && tot / (double)cnt > 1.0;
    }
    orc_impute_mean(X.data(), l, n);
    std::vector<double> Z(l * n), UtX(l * n), UtZ(l * n);
    for (size_t s = 0; s < l; ++s)
      for (size_t i = 0; i < n; ++i) {
        if (flip[s]) X[s * n + i] = 2.0 - X[s * n + i];
        Z[s * n + i] = X[s * n + i] * g_lmm.env[i];
      }
    gemma_hip_dgemm('N', 'N', l, n, n, 1.0, X.data(), n, g_lmm.U.data(), n, 0.0, UtX.data(), n);
    gemma_hip_dgemm('N', 'N', l, n, n, 1.0, Z.data(), n, g_lmm.U.data(), n, 0.0, UtZ.data(), n);
    orc_mvlmm_batch_gxe(g_lmm.cfg.a_mode, &g_lmm.mv_cfg, n, g_lmm.d, c + 1, g_lmm.eval.data(), g_lmm.Wt.data(), g_lmm.Yt.data(),
                        UtX.data(), UtZ.data(), l, g_lmm.mv_null.Vg_mle, g_lmm.mv_null.Ve_mle, g_lmm.mv_null.B_mle, out);
    const size_t d = g_lmm.d, stride = d + 3 * (d * (d + 1) / 2) + 3;
    for (size_t s = 0; s < l; ++s)
      if (flip[s])
        for (size_t i = 0; i < d; ++i) out[s * stride + i] = -out[s * stride + i];
    return GEMMA_HIP_OK;
  }
  orc_impute_mean(X.data(), l, n);
  std::vector<double> UtX(l * n);
  gemma_hip_dgemm('N', 'N', l, n, n, 1.0, X.data(), n, g_lmm.U.data(), n, 0.0, UtX.data(), n);
  orc_mvlmm_batch(g_lmm.cfg.a_mode, &g_lmm.mv_cfg, n, g_lmm.d, c, g_lmm.eval.data(), g_lmm.Wt.data(), g_lmm.Yt.data(),
                  UtX.data(), l, g_lmm.mv_null.Vg_mle, g_lmm.mv_null.Ve_mle, g_lmm.mv_null.B_mle, g_lmm.mv_null.logl_mle_H0, out);
  return GEMMA_HIP_OK;
}

// ---- -lm over orc_lm_batch ----------------------------------------------------------------------------------------
int gemma_hip_lm_setup(int a_mode, size_t n, size_t c, const double *W, const double *y) {
  g_lm.on = true;
  g_lm.a_mode = a_mode;
  g_lm.n = n;
  g_lm.c = c;
  g_lm.W.assign(W, W + n * c);
  g_lm.y.assign(y, y + n);
  std::vector<double> A(c * 2 * c, 0.0); // (W^T W)^-1 by Gauss-Jordan
  for (size_t a = 0; a < c; ++a) {
    for (size_t b = 0; b < c; ++b)
      for (size_t i = 0; i < n; ++i) A[a * 2 * c + b] += W[i * c + a] * W[i * c + b];
    A[a * 2 * c + c + a] = 1.0;
  }
  for (size_t p = 0; p < c; ++p) {
    size_t piv = p;
    for (size_t r = p + 1; r < c; ++r)
      if (std::fabs(A[r * 2 * c + p]) > std::fabs(A[piv * 2 * c + p])) piv = r;
    for (size_t j = 0; j < 2 * c; ++j) std::swap(A[p * 2 * c + j], A[piv * 2 * c + j]);
    const double d = A[p * 2 * c + p];
    for (size_t j = 0; j < 2 * c; ++j) A[p * 2 * c + j] /= d;
    for (size_t r = 0; r < c; ++r) {
      if (r == p) continue;
      const double f = A[r * 2 * c + p];
      for (size_t j = 0; j < 2 * c; ++j) A[r * 2 * c + j] -= f * A[p * 2 * c + j];
    }
  }
  g_lm.WtWi.assign(c * c, 0.0);
  for (size_t a = 0; a < c; ++a)
    for (size_t b = 0; b < c; ++b) g_lm.WtWi[a * c + b] = A[a * 2 * c + c + b];
  g_lmm.ind.clear();
  return GEMMA_HIP_OK;
}
int gemma_hip_lm_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  if (!g_lm.on) return fail(GEMMA_HIP_ESTATE, "lm_batch before lm_setup");
  const size_t n = g_lm.n;
  std::vector<double> X;
  if (kind == GEMMA_GENO_PLINK_2BIT)
    decode(kind, geno, l, ld, g_lmm.ind.empty() ? nullptr : g_lmm.ind.data(), g_lmm.ind.empty() ? n : g_lmm.ind.size(), n, X);
  else
    decode(kind, geno, l, ld, nullptr, n, n, X);
  orc_impute_mean(X.data(), l, n);
  orc_lm_batch(g_lm.a_mode, n, g_lm.c, g_lm.W.data(), g_lm.WtWi.data(), g_lm.y.data(), X.data(), l,
               reinterpret_cast<orc_sumstat *>(out));
  return GEMMA_HIP_OK;
}
int gemma_hip_lm_finish(void) {
  g_lm.on = false;
  return GEMMA_HIP_OK;
}

// referenced by inline members of class LMM the driver does not call (the linker still wants them with -O0)
// LMM::AnalyzeGene: rows of Y are phenotypes, the Uty slot of lmm_setup holds the rotated tested variable
int gemma_hip_lmm_gene_batch(const double *Y, size_t l, size_t ld, gemma_sumstat *out) {
  if (!g_lmm.on) return fail(GEMMA_HIP_ESTATE, "lmm_gene_batch before lmm_setup");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  std::vector<double> UtY(l * n);
  gemma_hip_dgemm('N', 'N', l, n, n, 1.0, Y, ld, g_lmm.U.data(), n, 0.0, UtY.data(), n); // row g = (U^T y_g)^T
  orc_gene_batch(g_lmm.cfg.a_mode, n, c, g_lmm.eval.data(), g_lmm.UtW.data(), g_lmm.Uty.data(), UtY.data(), l,
                 g_lmm.cfg.l_min, g_lmm.cfg.l_max, g_lmm.cfg.n_region, reinterpret_cast<orc_sumstat *>(out));
  return GEMMA_HIP_OK;
}

// ---- -gxe over orc_gxe_batch (feeder part as in LMM::AnalyzePlinkGXE, src/lmm.cpp:2490-2538) -------------------------
int gemma_hip_lmm_set_env(const double *env) {
  if (!g_lmm.on) return fail(GEMMA_HIP_ESTATE, "lmm_set_env before lmm_setup");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  g_lmm.env.assign(env, env + n);
  std::vector<double> Ute(n, 0.0);
  for (size_t i = 0; i < n; ++i)
    for (size_t k = 0; k < n; ++k) Ute[k] += g_lmm.U[i * n + k] * env[i];
  g_lmm.UtWe.assign(n * (c + 1), 0.0);
  for (size_t i = 0; i < n; ++i) {
    for (size_t a = 0; a < c; ++a) g_lmm.UtWe[i * (c + 1) + a] = g_lmm.UtW[i * c + a];
    g_lmm.UtWe[i * (c + 1) + c] = Ute[i];
  }
  return GEMMA_HIP_OK;
}
int gemma_hip_lmm_gxe_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  if (!g_lmm.on || g_lmm.env.empty()) return fail(GEMMA_HIP_ESTATE, "lmm_gxe_batch before lmm_set_env");
  const size_t n = g_lmm.cfg.n, c = g_lmm.cfg.n_cvt;
  std::vector<double> X;
  if (kind == GEMMA_GENO_PLINK_2BIT)
    decode(kind, geno, l, ld, g_lmm.ind.empty() ? nullptr : g_lmm.ind.data(), g_lmm.ind.empty() ? n : g_lmm.ind.size(), n, X);
  else
    decode(kind, geno, l, ld, nullptr, n, n, X);
  std::vector<int> flip(l, 0);
  for (size_t s = 0; s < l; ++s) { // x_mean over the non-missing calls decides the 2 - x recoding (:2519-2536)
    double tot = 0.0;
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i)
      if (X[s * n + i] == X[s * n + i]) { tot += X[s * n + i]; ++cnt; }
    flip[s] = cnt && tot / (double)cnt > 1.0;
  }
  orc_impute_mean(X.data(), l, n);
  std::vector<double> Z(l * n), UtX(l * n), UtZ(l * n);
  for (size_t s = 0; s < l; ++s)
    for (size_t i = 0; i < n; ++i) {
      if (flip[s]) X[s * n + i] = 2.0 - X[s * n + i];
      Z[s * n + i] = X[s * n + i] * g_lmm.env[i];
    }
  gemma_hip_dgemm('N', 'N', l, n, n, 1.0, X.data(), n, g_lmm.U.data(), n, 0.0, UtX.data(), n);
  gemma_hip_dgemm('N', 'N', l, n, n, 1.0, Z.data(), n, g_lmm.U.data(), n, 0.0, UtZ.data(), n);
  orc_gxe_batch(g_lmm.cfg.a_mode, n, c, g_lmm.eval.data(), g_lmm.UtWe.data(), g_lmm.Uty.data(), UtX.data(), UtZ.data(),
                flip.data(), l, g_lmm.cfg.l_min, g_lmm.cfg.l_max, g_lmm.cfg.n_region, g_lmm.cfg.l_mle_null,
                reinterpret_cast<orc_sumstat *>(out));
  return GEMMA_HIP_OK;
}

// ---- the device-resident chain and the communicator, on host memory (same semantics as the product's) ----
namespace {
struct Kept {
  std::vector<double> K, U, eval;
  size_t K_n = 0, n = 0;
  double trace = 0.0;
} g_kept;
struct DoubleComm {
  int rank = 0, world = 1;
  bool active = false;
  gemma_hip::ShmTransport tr;
} g_comm;
} // namespace

int gemma_hip_kin_end_keep(size_t *ns_used, int allreduce) {
  const size_t n = g_kin.n;
  double ns = (double)g_kin.ns;
  if (allreduce && g_comm.active && g_comm.world > 1) {
    g_comm.tr.allreduce_host(g_kin.K.data(), n * n);
    g_comm.tr.allreduce_host(&ns, 1);
  }
  g_kept.K.resize(n * n);
  for (size_t i = 0; i < n * n; ++i) g_kept.K[i] = g_kin.K[i] / ns;
  g_kept.K_n = n;
  if (ns_used) *ns_used = (size_t)(ns + 0.5);
  return GEMMA_HIP_OK;
}
int gemma_hip_kept_K_get(double *K) {
  if (!g_kept.K_n) return fail(GEMMA_HIP_ESTATE, "kept_K_get: no kept K");
  std::copy(g_kept.K.begin(), g_kept.K.end(), K);
  return GEMMA_HIP_OK;
}
static int kept_eigh_of(std::vector<double> &G, size_t n, double *eval, double *trace_G) {
  g_kept.U.assign(n * n, 0.0);
  g_kept.eval.assign(n, 0.0);
  double tr = 0.0;
  const int rc = gemma_hip_eigh(G.data(), n, g_kept.U.data(), g_kept.eval.data(), &tr);
  if (rc) return rc;
  g_kept.n = n;
  g_kept.trace = tr;
  if (trace_G) *trace_G = tr;
  if (eval) std::copy(g_kept.eval.begin(), g_kept.eval.end(), eval);
  return GEMMA_HIP_OK;
}
int gemma_hip_eigh_kept_K(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G) {
  if (!g_kept.K_n || ni_total != g_kept.K_n) return fail(GEMMA_HIP_ESTATE, "eigh_kept_K: no matching kept K");
  std::vector<size_t> map;
  for (size_t i = 0; i < ni_total; ++i)
    if (!indicator_idv || indicator_idv[i]) map.push_back(i);
  const size_t n = map.size();
  std::vector<double> G(n * n);
  for (size_t r = 0; r < n; ++r)
    for (size_t c = 0; c < n; ++c) G[r * n + c] = g_kept.K[map[r] * ni_total + map[c]];
  orc_CenterMatrix(G.data(), n);
  return kept_eigh_of(G, n, eval, trace_G);
}
// the collective form: on this CPU double every rank decomposes its (identical, all-reduced) kept K itself -- LAPACK on the host
// is deterministic, so all ranks end with the same (U, eval), which is the entry point's contract
int gemma_hip_eigh_kept_K_sharded(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G) {
  return gemma_hip_eigh_kept_K(indicator_idv, ni_total, eval, trace_G);
}
int gemma_hip_eigh_keep(const double *G, size_t n, double *eval, double *trace_G) {
  std::vector<double> Gc(G, G + n * n);
  return kept_eigh_of(Gc, n, eval, trace_G);
}
int gemma_hip_kept_n(size_t *n) {
  if (n) *n = g_kept.n;
  return GEMMA_HIP_OK;
}
int gemma_hip_kept_bcast(int root, double *trace_G) {
  if (g_comm.active && g_comm.world > 1) {
    double hdr[2] = {(double)g_kept.n, g_kept.trace};
    g_comm.tr.bcast_host(hdr, 16, root);
    const size_t n = (size_t)(hdr[0] + 0.5);
    if (g_comm.rank != root) {
      g_kept.U.assign(n * n, 0.0);
      g_kept.eval.assign(n, 0.0);
      g_kept.n = n;
      g_kept.trace = hdr[1];
    }
    g_comm.tr.bcast_host(g_kept.U.data(), n * n * 8, root);
    g_comm.tr.bcast_host(g_kept.eval.data(), n * 8, root);
  }
  if (trace_G) *trace_G = g_kept.trace;
  return GEMMA_HIP_OK;
}
int gemma_hip_kept_U_get(double *U, double *eval) {
  if (!g_kept.n) return fail(GEMMA_HIP_ESTATE, "kept_U_get: no kept U");
  if (U) std::copy(g_kept.U.begin(), g_kept.U.end(), U);
  if (eval) std::copy(g_kept.eval.begin(), g_kept.eval.end(), eval);
  return GEMMA_HIP_OK;
}
int gemma_hip_calc_utx_kept(const double *X, size_t n, size_t m, double *UtX) {
  if (!g_kept.n || n != g_kept.n) return fail(GEMMA_HIP_ESTATE, "calc_utx_kept: no matching kept U");
  return gemma_hip_dgemm('T', 'N', n, m, n, 1.0, g_kept.U.data(), n, X, m, 0.0, UtX, m);
}
int gemma_hip_lmm_setup_kept(const gemma_lmm_cfg *cfg, const double *UtW, const double *Uty) {
  if (!g_kept.n || cfg->n != g_kept.n) return fail(GEMMA_HIP_ESTATE, "lmm_setup_kept: no matching kept U");
  return gemma_hip_lmm_setup(cfg, g_kept.U.data(), g_kept.eval.data(), UtW, Uty);
}
int gemma_hip_kept_release(void) {
  g_kept = Kept();
  return GEMMA_HIP_OK;
}

int gemma_hip_comm_unique_id(void *id) {
  gemma_hip::ShmTransport::make_id(id);
  return GEMMA_HIP_OK;
}
int gemma_hip_comm_init(const void *id, int rank, int world) {
  g_comm.rank = rank;
  g_comm.world = world;
  g_comm.active = true;
  if (world > 1) {
    std::string err;
    if (!g_comm.tr.open(id, rank, world, err)) return fail(GEMMA_HIP_ERUNTIME, err.c_str());
  }
  return GEMMA_HIP_OK;
}
int gemma_hip_comm_info(int *rank, int *world, int *transport) {
  if (rank) *rank = g_comm.rank;
  if (world) *world = g_comm.world;
  if (transport) *transport = g_comm.world > 1 ? 2 : 0;
  return GEMMA_HIP_OK;
}
int gemma_hip_comm_finalize(void) {
  if (g_comm.tr.is_open()) g_comm.tr.close_segment();
  g_comm.active = false;
  g_comm.rank = 0;
  g_comm.world = 1;
  return GEMMA_HIP_OK;
}
int gemma_hip_comm_selftest(void *) { return GEMMA_HIP_OK; }
int gemma_hip_comm_stats(gemma_comm_stats *out) {
  if (out) memset(out, 0, sizeof *out);
  return GEMMA_HIP_OK;
}
// the eigensolver's device workspace: nothing to reserve on the host double
int gemma_hip_eigh_reserve(size_t) { return GEMMA_HIP_OK; }
int gemma_hip_eigh_release(size_t *bytes_freed) {
  if (bytes_freed) *bytes_freed = 0;
  return GEMMA_HIP_OK;
}
}
