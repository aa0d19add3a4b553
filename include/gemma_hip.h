/*
 * gemma_hip.h -- C ABI of the MI355X (gfx950) implementation of GEMMA's
 * kinship + univariate-LMM hot path.
 *
 * GEMMA has no plugin/FFI layer; the boundary is cut at the internal C++
 * waists listed in SURVEY.md section 8(b).  Every entry point below names the
 * reference interface it replaces (file:line relative to the GEMMA tree).
 * INTEGRATION.md shows the few lines a GEMMA maintainer adds at each site.
 *
 * Conventions
 *  - plain pointers and sizes only; all matrices are row-major with an explicit
 *    leading dimension (maps 1:1 onto gsl_matrix{data,tda});
 *  - every function returns 0 on success or a GEMMA_HIP_E* code;
 *    gemma_hip_last_error() holds the detail text of the last failure;
 *  - entry points without suffix take HOST pointers (the library stages through
 *    device memory and never frees caller memory); the `_d` twins take DEVICE
 *    pointers plus a hipStream_t (passed as void*, NULL = the null stream) and
 *    are asynchronous on that stream;
 *  - one calling host thread (GEMMA is single-threaded); one process per GPU.
 *    Multi-GPU = one process per device, SNP blocks sharded by the caller and
 *    (U, eval, UtW, Uty) broadcast once into every rank's buffers (RCCL) before
 *    gemma_hip_lmm_setup_d;
 *  - there is NO CPU fallback: without a usable gfx950 device every call
 *    returns GEMMA_HIP_ENODEV.
 */
#ifndef GEMMA_HIP_H
#define GEMMA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: gemma_mvlmm_null sized for 8 phenotypes / 11 covariates, gemma_mvlmm_opt gained `gxe` (round 4); 3: gemma_mvlmm_opt gained
 * `crt` (round 3); 2: the round-2 entry points */
#define GEMMA_HIP_ABI_VERSION 4

enum {
  GEMMA_HIP_OK = 0,
  GEMMA_HIP_EINVAL = 1,   /* bad argument / shape ("Range error in dgemm", src/fastblas.cpp:207) */
  GEMMA_HIP_ENODEV = 2,   /* no gfx950 device / not initialised */
  GEMMA_HIP_ENOMEM = 3,   /* device allocation failed */
  GEMMA_HIP_ERUNTIME = 4, /* HIP runtime / kernel failure */
  GEMMA_HIP_ESTATE = 5,   /* call sequence violated (e.g. lmm_batch before lmm_setup) */
  GEMMA_HIP_ENOCONV = 6   /* eigensolver did not converge (INFO != 0, src/lapack.cpp:213,224) */
};

/* == class SUMSTAT, src/param.h:54-66 (8 doubles per analysed SNP, in SNP order) */
typedef struct {
  double beta, se, lambda_remle, lambda_mle, p_wald, p_lrt, p_score, logl_H1;
} gemma_sumstat;

/* genotype block encodings accepted by kin_add / lmm_batch */
enum {
  /* fp64, SNP-major: l rows of ld doubles (ld >= n); NaN = missing.  Rows hold exactly
   * the individuals the consumer works on (analysed ones for lmm_batch, all for kin_add). */
  GEMMA_GENO_F64_SNP_MAJOR = 0,
  /* PLINK .bed payload, SNP-major: l rows of ld bytes (ld >= ceil(ni_total/4)), 4 individuals
   * per byte, low bits first, codes 00->2 01->missing(bit0=1,bit1=0) 10->1 11->0 exactly as
   * src/lmm.cpp:1797-1812.  Individuals are dropped on device with the indicator given to
   * gemma_hip_lmm_set_indicator (lmm_batch only; kin_add always uses all ni_total). */
  GEMMA_GENO_PLINK_2BIT = 1,
  /* fp64, individual-major: n rows of ld doubles, SNP j = strided column j (ld >= l) -- the
   * reference's own Xlarge layout (src/lmm.cpp:1502,1635; src/gemma_io.cpp:1439,1546).
   * Missing values must already be imputed by the caller (as the reference does before the
   * copy into Xlarge); for kin_add the columns must already be centred/scaled. */
  GEMMA_GENO_F64_IDV_MAJOR = 2
};

/* ---- lifetime ---------------------------------------------------------- */
int gemma_hip_init(int device /* -1: keep the current HIP device */, int verbose);
void gemma_hip_shutdown(void);
int gemma_hip_abi_version(void);
const char *gemma_hip_strerror(int code);
const char *gemma_hip_last_error(void);
/* name[len], number of CUs, HBM bytes of the device in use */
int gemma_hip_device_info(char *name, size_t len, int *n_cu, size_t *hbm_bytes);

/* ---- B2: GEMM ---------------------------------------------------------- */
/* replaces fast_dgemm / fast_eigen_dgemm, src/fastblas.h:34-39 (-> cblas_dgemm,
 * src/fastblas.cpp:202-204) and the raw fast_cblas_dgemm, src/fastblas.cpp:66-170:
 * C(MxN) = alpha*op(A)*op(B) + beta*C, row-major, ta/tb in {'N','T'}.
 * fp64 MFMA (v_mfma_f64_16x16x4_f64), LDS-tiled. */
int gemma_hip_dgemm(char ta, char tb, size_t M, size_t N, size_t K, double alpha, const double *A,
                    size_t lda, const double *B, size_t ldb, double beta, double *C, size_t ldc);
int gemma_hip_dgemm_d(char ta, char tb, size_t M, size_t N, size_t K, double alpha,
                      const double *A_d, size_t lda, const double *B_d, size_t ldb, double beta,
                      double *C_d, size_t ldc, void *stream);

/* ---- B1: streaming kinship --------------------------------------------- */
/* replaces the body of BimbamKin / PlinkKin, src/gemma_io.cpp:1418-1597 / :1599-1738, called
 * from PARAM::CalcKin src/param.cpp:1300-1321: per SNP over all n_total individuals mean over
 * non-missing, impute, centre, (k_mode 2: scale by 1/sqrt(var), var as :1511-1514),
 * K += Xb Xb^T per block, finally K /= ns_used (:1570) and symmetric fill (:1724-1729). */
int gemma_hip_kin_begin(size_t n_total, int k_mode /* 1 centred, 2 standardised */);
int gemma_hip_kin_add(int geno_kind, const void *geno, size_t l, size_t ld);
int gemma_hip_kin_add_d(int geno_kind, const void *geno_d, size_t l, size_t ld, void *stream);
int gemma_hip_kin_end(double *K /* n_total^2, row-major, full symmetric */, size_t *ns_used);
int gemma_hip_kin_end_d(double *K_d, size_t *ns_used, void *stream);

/* leave-one-chromosome-out kinship (-loco, src/param.cpp:52-66,497-500): given the kinship of ALL
 * analysed SNPs (ns_all of them) and of the SNPs on one chromosome (ns_chr), both from kin_begin/add/end_d,
 * overwrite K_chr_d with the kinship of the remaining SNPs: (ns_all K_all - ns_chr K_chr)/(ns_all - ns_chr). */
int gemma_hip_kin_loco_d(const double *K_all_d, size_t ns_all, double *K_chr_d, size_t ns_chr, size_t n,
                         void *stream);

/* ---- first-pass SNP QC ------------------------------------ */
/* The per-SNP filters of ReadFile_geno (src/gemma_io.cpp:753-853; kind F64_SNP_MAJOR, rows over all ni_total
 * individuals, NaN = "NA") and ReadFile_bed (:942-1049; kind PLINK_2BIT), statistics over the analysed
 * individuals only: missingness > miss_level, maf outside [maf_level, 1-maf_level] (skipped when maf_level == -1),
 * non-polymorphic, HWE exact test p < hwe_level (0 = off), r2 with the covariates > r2_level (only when
 * n_cvt > 1).  W is the n x n_cvt covariate matrix of the analysed individuals (host, row-major).
 * indicator_idv may be NULL when ni_total == n.  Outputs (host): indicator_snp[l], maf[l], n_miss[l]
 * (the latter two may be NULL). */
typedef struct {
  double maf_level;  /* 0.01  (src/param.cpp:94-107) */
  double miss_level; /* 0.05 */
  double hwe_level;  /* 0 */
  double r2_level;   /* 0.9999 */
} gemma_qc_cfg;
int gemma_hip_snp_qc(int geno_kind, const void *geno, size_t l, size_t ld, const int *indicator_idv,
                     size_t ni_total, const double *W, size_t n, size_t n_cvt, const gemma_qc_cfg *cfg,
                     int *indicator_snp, double *maf, size_t *n_miss);

/* ---- B3: centring + eigendecomposition --------------------------------- */
/* CenterMatrix(gsl_matrix*), src/mathfunc.cpp:147-177 (in place) */
int gemma_hip_center(double *G, size_t n);
int gemma_hip_center_d(double *G_d, size_t n, void *stream);
/* EigenDecomp_Zeroed, src/lapack.cpp:260-291 (-> lapack_eigen_symmv/dsyevr_, :149-236):
 * G (destroyed) -> U (row-major, eigenvector k = column k), eval ascending with values
 * < 1e-10 set to 0; *trace_G = mean(eval) (the function's return value, :277). */
int gemma_hip_eigh(double *G, size_t n, double *U, double *eval, double *trace_G);
int gemma_hip_eigh_d(double *G_d, size_t n, double *U_d, double *eval_d, double *trace_G,
                     void *stream);
/* The solver's device workspace is ~5 n^2 doubles in thirty-odd buffers, and hipMalloc / hipFree of such sizes cost 25-50 ms per GB
 * (0.1-0.2 s of a 1.9 s solve at n = 20 000, 1-2 s of 19 s at n = 50 000).  gemma_hip_eigh_reserve(n) allocates the workspace of order n
 * ahead of the solve (a caller that knows its n does this while it reads its files) and from then on every solve leaves its buffers in a
 * pool for the next one (LOCO: one decomposition per chromosome); gemma_hip_eigh_release hands the idle buffers back (lmm_setup* does
 * so by itself when the pool holds more than a quarter of the device).  GEMMA_HIP_EIGH_CACHE=1 keeps every solve's buffers without a
 * reserve call.  Without either the solver allocates and frees per call, as before.  gemma_hip_eigh_reserve touches nothing but the solver's
 * own pool: after gemma_hip_init it is the ONE entry point that may run on a second thread beside another call of the library (the file
 * driver issues it while the first pass reads the genotype file); join it before the first gemma_hip_eigh* call. */
int gemma_hip_eigh_reserve(size_t n);
int gemma_hip_eigh_release(size_t *bytes_freed /* may be NULL */);
/* The same as a COLLECTIVE over the library's communicator (gemma_hip_comm_init): every rank passes the same G and
 * receives the same (U, eval).  The reduction and the divide & conquer run on every rank (same code, same bits -- checked),
 * the two back-transformations are shared out by eigenvector (rows of Z^T) and the slices exchanged once; with one rank, or
 * GEMMA_HIP_EIGH_SHARD=0, it is gemma_hip_eigh_d. */
int gemma_hip_eigh_sharded_d(double *G_d, size_t n, double *U_d, double *eval_d, double *trace_G, void *stream);
/* CalcUtX(U,X,UtX), src/mathfunc.cpp:504-506: UtX (n x m) = U^T X (X n x m row-major) */
int gemma_hip_calc_utx(const double *U, const double *X, size_t n, size_t m, double *UtX);

/* ---- B4: per-batch association ------------------------------------------ */
/* the state batch_compute (src/lmm.cpp:1513-1564) reads from class LMM, src/lmm.h:53-74 */
typedef struct {
  int a_mode;          /* 1 Wald, 2 LRT, 3 score, 4 all, 9 (src/gemma.h:39-43) */
  size_t n;            /* ni_test */
  size_t n_cvt;        /* covariates incl. intercept (>= 1) */
  double l_min, l_max; /* 1e-5, 1e5 (src/param.cpp:94-107) */
  size_t n_region;     /* 10 */
  double l_mle_null;   /* null-model ML lambda (score test, src/lmm.cpp:1542) */
  double logl_mle_H0;  /* null-model ML log-likelihood (LRT, src/lmm.cpp:1553) */
  int plink_nan_rule;  /* 1: AnalyzePlink's handling of a failed lambda search (src/lmm.cpp:1870-1884) */
} gemma_lmm_cfg;

/* Null model (calc_null = true): CalcLambda('L'/'R', eval, UtW, Uty, ...) src/lmm.cpp:2143-2180 as
 * called at src/gemma.cpp:2711,2734, plus CalcPve (src/lmm.cpp:2183-2205) and the vg/ve part of
 * CalcLmmVgVeBeta (:2253-2259).  Host pointers.  out8 = { l_mle_null, logl_mle_H0, l_remle_null,
 * logl_remle_H0, pve, pve_se, vg_remle, ve_remle }. */
int gemma_hip_lmm_null(size_t n, size_t n_cvt, const double *eval, const double *UtW,
                       const double *Uty, double l_min, double l_max, size_t n_region,
                       double trace_G, double *out8);

/* uploads U (n x n row-major, eigenvectors in columns), eval, UtW (n x n_cvt row-major), Uty */
int gemma_hip_lmm_setup(const gemma_lmm_cfg *cfg, const double *U, const double *eval,
                        const double *UtW, const double *Uty);
/* same with device-resident inputs (borrowed until lmm_finish): the multi-GPU path, after the
 * single RCCL broadcast of (U, eval, UtW, Uty) */
int gemma_hip_lmm_setup_d(const gemma_lmm_cfg *cfg, const double *U_d, const double *eval_d,
                          const double *UtW_d, const double *Uty_d, void *stream);
/* for GEMMA_GENO_PLINK_2BIT: indicator_idv over ni_total individuals (sum == cfg.n);
 * NULL/0 resets to "all individuals analysed" */
int gemma_hip_lmm_set_indicator(const int *indicator_idv, size_t ni_total);
/* one block of l SNPs: mean-impute (src/lmm.cpp:1590-1618 / :1819-1827), UtX = U^T X (:1521),
 * then per SNP CalcUab, CalcRLScore, CalcLambda('R')+CalcRLWald, CalcLambda('L')+LRT (:1526-1562).
 * out[l] in SNP order. l is not limited to LMM_BATCH_SIZE.
 * U^T X: fp64 MFMA GEMM for real-valued input; for hard calls (GEMMA_GENO_PLINK_2BIT, or fp64 rows holding only
 * 0/1/2 and one missing / imputed value -- detected per batch) the same product as 2 D int8 MFMA products of
 * {genotype, missing mask} with D balanced base-256 digits of U, accumulated exactly in int32.  PRECISION OF THE OPERAND (round 6):
 * every column of U is scaled by its exact maximum and rounded at 1.01 * 2^(-8 D) of it.  D = 7 below n = 16384: 2^-56 of the column
 * maximum, below an fp64 entry's own rounding in the column's top binades, and the product is closer to the exact dot products than an
 * fp64 GEMM (which rounds every partial sum).  From n = 16384 up D = 6: U is ROUNDED to 2^-48 of each column's maximum -- narrower than
 * the reference's fp64 operand (2^-53 per entry); measured against long-double products the rms error of U^T x is 4.6 x that of the fp64
 * MFMA GEMM, the maximum 4.4 x, eight orders below the 1e-6 bar on the statistics.  GEMMA_HIP_I8_FORM=7g6m is the strict
 * form -- seven digits for the genotype product, the mask product (whose term is sqrt(n / missing calls) smaller) on the upper six:
 * rms and maximum error a quarter / an eighth of the fp64 GEMM's for 10/9 of the matrix work; GEMMA_HIP_I8_DIGITS=7 seven digits for both at
 * any n (14 products: 7/6 of the time), GEMMA_HIP_UTX_I8=0 selects the fp64 GEMM always.  The _d form is asynchronous on its stream for
 * GEMMA_GENO_PLINK_2BIT; for fp64 input it synchronises the stream once per call (the hard-call verdict is read back). */
int gemma_hip_lmm_batch(int geno_kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out);
int gemma_hip_lmm_batch_d(int geno_kind, const void *geno_d, size_t l, size_t ld,
                          gemma_sumstat *out_d, void *stream);
/* gemma_hip_lmm_batch_d with TWO BLOCKS IN FLIGHT on a CU partition (round 5; the loop of src/lmm.cpp:1526-1562 has no dependency
 * between blocks except AnalyzePlink's beta / se carry, which stays in block order): ingest, records and the int8 product of this
 * block run on 192 of the 256 CUs while the digit combine and the per-SNP stage of the PREVIOUS block run on the other 64 (two of the
 * eight CUs of every shader engine; GEMMA_HIP_PIPE_CUS=32|64|...; 0 = two plain streams).  The call returns when the work is QUEUED:
 * out_d of a block is complete only after gemma_hip_lmm_pipe_flush(stream) has ordered `stream` behind it (or after any other
 * batch entry point, which flushes first); geno_d may be overwritten once work queued on `stream` AFTER the next call (or the
 * flush) runs.  Records are the ones gemma_hip_lmm_batch_d writes, bit for bit.  PLINK 2-bit blocks on the records kernel only;
 * every other input takes gemma_hip_lmm_batch_d. */
int gemma_hip_lmm_batch_pipe_d(int geno_kind, const void *geno_d, size_t l, size_t ld, gemma_sumstat *out_d, void *stream);
int gemma_hip_lmm_pipe_flush(void *stream);
/* The same for a STREAM of host blocks, pipelined (pinned, double-buffered H2D; the feeder this replaces is the
 * per-SNP read loop of src/lmm.cpp:1776-1827): submit copies the block into one of two pinned staging slots and queues its
 * H2D copy (copy stream), the batch (compute stream, after the copy) and the D2H of its SUMSTAT records; collect waits for
 * the OLDEST submitted block and hands its records over.  At most two blocks in flight: submit(k + 1) before collect(k) moves
 * block k + 1 across PCIe while block k computes.  The caller's buffer is free again when submit returns. */
int gemma_hip_lmm_batch_submit(int geno_kind, const void *geno, size_t l, size_t ld);
int gemma_hip_lmm_batch_collect(gemma_sumstat *out /* room for the block's l records */, size_t *l /* may be NULL */);
/* LMM::AnalyzeGene (src/lmm.cpp:1365-1471; `-gene`): every row of Y (l x ld, row-major, ld >= cfg.n) is a PHENOTYPE
 * (a gene's expression over the analysed individuals); the tested variable is the one fixed vector whose rotation
 * U^T x was handed to gemma_hip_lmm_setup in the Uty slot.  Per row: U^T y_g (:1415), the row's own null ML fit
 * (l_H0, logl_H0, :1424-1427), CalcRLScore at l_H0, CalcLambda('R')+CalcRLWald, CalcLambda('L')+LRT against logl_H0
 * (:1434-1450).  l_mle_null / logl_mle_H0 of the cfg are not used. */
int gemma_hip_lmm_gene_batch(const double *Y, size_t l, size_t ld, gemma_sumstat *out);
int gemma_hip_lmm_gene_batch_d(const double *Y_d, size_t l, size_t ld, gemma_sumstat *out_d, void *stream);
/* GXE variants, LMM::AnalyzeBimbamGXE / AnalyzePlinkGXE (src/lmm.cpp:2283-2608; `-gxe`): the covariates of SNP s are
 * [W, env, x_s] and the tested variable is x_s . env.  After lmm_setup: set_env(env over the cfg.n analysed individuals)
 * rotates env (:2308); gxe_batch takes SNP-major blocks (GEMMA_GENO_F64_SNP_MAJOR or GEMMA_GENO_PLINK_2BIT with the
 * lmm_set_indicator mapping), mean-imputes, recodes 2 - x when x_mean > 1 (beta changes sign, :2352-2354,:2403), and per
 * SNP runs the c+2-covariate null ML fit (a_mode 2/4), CalcRLScore at cfg.l_mle_null, REML + Wald, ML + LRT (:2376-2400). */
int gemma_hip_lmm_set_env(const double *env);
int gemma_hip_lmm_gxe_batch(int geno_kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out);
int gemma_hip_lmm_gxe_batch_d(int geno_kind, const void *geno_d, size_t l, size_t ld, gemma_sumstat *out_d, void *stream);
/* the second half of lmm_batch on a caller-supplied UtX (SNP-major l x ld_utx, device):
 * what remains of batch_compute after the fast_dgemm at src/lmm.cpp:1521 */
int gemma_hip_lmm_assoc_d(const double *UtX_d, size_t l, size_t ld_utx, gemma_sumstat *out_d,
                          void *stream);
/* releases the LMM state; reports GPU time spent in the UtX GEMM and in the per-SNP stage in
 * minutes, the unit of LMM::time_UtX / time_opt (src/lmm.cpp:1523,1556) */
int gemma_hip_lmm_finish(double *time_UtX_min, double *time_opt_min);

/* ---- multivariate LMM (-lmm 1..4 -n a b c ...) ------------------------------------ */
/* MVLMM::AnalyzeBimbam / AnalyzePlink, src/mvlmm.cpp:2972-3416 / :3418-3899, and with opt->gxe the interaction test of
 * AnalyzeBimbamGXE / AnalyzePlinkGXE (:3970-4414 / :4416-4870).  d phenotypes (1..GEMMA_MV_DMAX), n_cvt covariates with
 * n_cvt + 1 <= GEMMA_MV_CMAX rows of X (gxe: n_cvt + 3): register-resident kernels for d <= 5 with up to 3 covariates and d <= 3
 * with up to 6, one run-time-shaped kernel (slower) for every other shape and for gxe.  opt->crt = 1 applies the Edgeworth
 * correction (CalcCRT / PCRT, :2054-2358 / :2952-2970) as -crt does.
 * gemma_mvlmm_null holds what the null-model block (:3056-3208) leaves behind: V_g, V_e (d x d row-major, leading
 * dimension d), B (d x n_cvt) and the log-likelihood, for the REMLE and the MLE fit. */
#define GEMMA_MV_DMAX 8
#define GEMMA_MV_CMAX 12
typedef struct {
  double Vg_remle[GEMMA_MV_DMAX * GEMMA_MV_DMAX], Ve_remle[GEMMA_MV_DMAX * GEMMA_MV_DMAX], B_remle[GEMMA_MV_DMAX * GEMMA_MV_CMAX],
      logl_remle_H0;
  double Vg_mle[GEMMA_MV_DMAX * GEMMA_MV_DMAX], Ve_mle[GEMMA_MV_DMAX * GEMMA_MV_DMAX], B_mle[GEMMA_MV_DMAX * GEMMA_MV_CMAX], logl_mle_H0;
} gemma_mvlmm_null;
/* PARAM defaults (src/param.cpp:94-107): em_iter 10000, em_prec 1e-4, nr_iter 100, nr_prec 1e-4, p_nr 1e-3, crt 0 */
typedef struct {
  size_t em_iter, nr_iter;
  double em_prec, nr_prec, p_nr;
  size_t crt; /* PARAM::crt (-crt, src/gemma.cpp:1398-1399): 1 = PCRT on the SNPs that reach MphNR (:3302-3306,3329-3331,3349-3351) */
  size_t gxe; /* 1 (-gxe with several phenotypes, src/gemma.cpp:2840-2851): after gemma_hip_lmm_set_env; mvlmm_set then takes the null
               * fit of (W, env) -- gemma_hip_mvlmm_null called with U^T env appended to UtW as covariate n_cvt + 1 (:4046-4070) --
               * and mvlmm_batch tests x o env per SNP with (W, env, x) as covariates; SNP-major input only; beta changes sign with
               * the allele switch of :4232-4236 */
} gemma_mvlmm_opt;
/* The null block: MphInitial (:2763-2948; one univariate REML fit per trait, and for d > 4 one two-trait fit per pair),
 * MphEM + MphNR + MphCalcBeta for 'R', then for 'L' starting from the REMLE fit.  Host pointers: eval (n), UtW (n x
 * n_cvt row-major), UtY (n x d row-major) -- the arguments of AnalyzeBimbam.  Stand-alone (no lmm_setup needed). */
int gemma_hip_mvlmm_null(size_t n, size_t n_cvt, size_t d, const double *eval, const double *UtW, const double *UtY,
                         double l_min, double l_max, size_t n_region, const gemma_mvlmm_opt *opt, gemma_mvlmm_null *out);
/* After gemma_hip_lmm_setup (which holds U, eval, UtW, cfg.a_mode, cfg.n, cfg.n_cvt; its Uty slot is not read by this
 * path): the rotated phenotypes UtY (n x d row-major, host) and the null fit the per-SNP loop starts from (:3206-3208:
 * the MLE block; logl_mle_H0 is the LRT reference). */
int gemma_hip_mvlmm_set(size_t d, const double *UtY, const gemma_mvlmm_null *null_fit, const gemma_mvlmm_opt *opt);
/* One block of SNPs (:3218-3375), same genotype encodings / indicator mapping / imputation as lmm_batch.  Per SNP the
 * record MPHSUMSTAT (src/param.h:68-77) as doubles: beta[d], Vbeta[v], Vg[v], Ve[v], p_wald, p_lrt, p_score with
 * v = d (d + 1) / 2 (upper triangles, row by row) -- out holds l x (d + 3 v + 3) doubles. */
int gemma_hip_mvlmm_batch(int geno_kind, const void *geno, size_t l, size_t ld, double *out);
int gemma_hip_mvlmm_batch_d(int geno_kind, const void *geno_d, size_t l, size_t ld, double *out_d, void *stream);

/* ---- linear model without a random effect (-lm 1..4) ----------------------- */
/* LM::AnalyzeBimbam / AnalyzePlink, src/lm.cpp:382-640 (CalcvPv :224-263, LmCalcP :266-287): per SNP ordinary
 * regression of y on (W, x); a_mode 51 Wald, 52 LRT, 53 score, 54 all (src/gemma.h:39-43).  W (n x n_cvt) and y are
 * host arrays over the analysed individuals.  Blocks use the same genotype encodings, indicator
 * (gemma_hip_lmm_set_indicator) and mean imputation as lmm_batch; SUMSTAT carries beta, se (score se for a_mode 53),
 * p_wald, p_lrt, p_score, the lambdas are 0 and logl_H1 is -0.0 as in the reference. */
int gemma_hip_lm_setup(int a_mode, size_t n, size_t n_cvt, const double *W, const double *y);
int gemma_hip_lm_batch(int geno_kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out);
int gemma_hip_lm_batch_d(int geno_kind, const void *geno_d, size_t l, size_t ld, gemma_sumstat *out_d, void *stream);
int gemma_hip_lm_finish(void);

/* ---- the device-resident chain behind the host-pointer API ------------------------------ */
/* In a file-driven run K, the analysed block G and U cross PCIe five times (K out of kin_end, G into eigh, U out, U in
 * for CalcUtX, U in for lmm_setup: 8 n^2 bytes each).  These entry points keep them where they are produced.  One kept K
 * and one kept (U, eval) per process; lmm_finish does not release them, gemma_hip_kept_release does.
 *   kin_begin / kin_add ... -> kin_end_keep            K (ni_total^2) stays on the device; with a communicator of more than one
 *                                                       rank and allreduce != 0 the ranks' partial sums over THEIR SNPs are
 *                                                       all-reduced first (SNP-sharded kinship) and ns_used is the total
 *   kept_K_get                                          copy of the kept K for <o>.cXX.txt (PARAM::WriteMatrix)
 *   eigh_kept_K(indicator_idv)                          rows / columns of the analysed individuals (what ReadFile_kin keeps,
 *                                                       src/gemma_io.cpp:1205-1243), CenterMatrix, EigenDecomp_Zeroed -- all on the
 *                                                       device; U and eval stay there, eval and trace_G also come back
 *   eigh_keep(G)                                        the same from a host G (the -k file route): G is centred by the caller
 *   kept_bcast(root)                                    ONE ncclBroadcast of (U, eval) to every rank of the communicator
 *   calc_utx_kept / lmm_setup_kept                      CalcUtX and lmm_setup on the kept U
 *   kept_U_get                                          copy of U / eval for the -eigen artefacts */
int gemma_hip_kin_end_keep(size_t *ns_used, int allreduce);
int gemma_hip_kept_K_get(double *K /* ni_total^2 */);
int gemma_hip_eigh_kept_K(const int *indicator_idv /* NULL: all */, size_t ni_total, double *eval /* n_test, may be NULL */,
                          double *trace_G);
int gemma_hip_eigh_keep(const double *G /* n^2, already centred; not modified */, size_t n, double *eval /* may be NULL */,
                        double *trace_G);
int gemma_hip_kept_n(size_t *n /* order of the kept U, 0 = none */);
/* gemma_hip_eigh_kept_K as a collective (every rank holds the same kept K after the all-reduce of kin_end_keep): every rank ends
 * with the same kept (U, eval); gemma_hip_kept_bcast is then not needed */
int gemma_hip_eigh_kept_K_sharded(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G);
int gemma_hip_kept_bcast(int root, double *trace_G /* out on every rank: mean(eval) of the root's decomposition; may be NULL */);
int gemma_hip_kept_U_get(double *U /* n^2, may be NULL */, double *eval /* n, may be NULL */);
int gemma_hip_calc_utx_kept(const double *X, size_t n, size_t m, double *UtX);
int gemma_hip_lmm_setup_kept(const gemma_lmm_cfg *cfg, const double *UtW, const double *Uty);
int gemma_hip_kept_release(void);

/* ---- multi-GPU: RCCL over xGMI, one process per GPU ------------------------------------------ */
/* SNPs are independent (src/lmm.cpp:1513-1514); the shared state is read-only.  Two collectives exist on the path: ONE
 * broadcast of (U, eval) from the rank that ran the eigensolver, and -- for a SNP-sharded kinship -- ONE all-reduce of the
 * n^2 partial sums.  Both are issued directly on RCCL (ncclBroadcast / ncclAllReduce, librccl bound with dlopen when a
 * communicator of more than one rank is created).  Bootstrap: rank 0 calls comm_unique_id (ncclGetUniqueId), the host
 * program ships the 128 bytes to the other ranks, every rank calls comm_init with its device current (ncclCommInitRank).
 * GEMMA_HIP_COMM=shm in the environment selects a host shared-memory transport instead: a TEST hook for boxes with one
 * GPU (RCCL refuses two ranks on one device); it is never chosen implicitly. */
#define GEMMA_HIP_COMM_ID_BYTES 128
int gemma_hip_comm_unique_id(void *id /* GEMMA_HIP_COMM_ID_BYTES */);
int gemma_hip_comm_init(const void *id /* may be NULL when world == 1 */, int rank, int world);
int gemma_hip_comm_info(int *rank, int *world, int *transport /* 0 none, 1 RCCL, 2 shm test transport */);
int gemma_hip_comm_bcast_d(void *buf_d, size_t bytes, int root, void *stream);
int gemma_hip_comm_allreduce_sum_d(double *buf_d, size_t count, void *stream);
int gemma_hip_comm_finalize(void);
/* The first contact of a new communicator: ONE KiB through both collectives (an all-reduce of rank + 1, a broadcast from rank 0 and one
 * from the last rank; every value checked on every rank) before anything n^2 is trusted to them.  Synchronous; a caller that cannot
 * afford a hang runs it under its own wall-clock deadline (bench.py: a helper thread).  Every collective above is cut into pieces of at
 * most 1 GiB issued back to back on the caller's stream.  GEMMA_HIP_COMM_FAIL=init|selftest|allreduce|bcast|allreduce_large|bcast_large
 * makes that entry point fail on every rank (the *_large forms only from 1 MiB up, i.e. after a passed self-test): failure injection
 * for the callers' fall-backs. */
int gemma_hip_comm_selftest(void *stream);
/* what the communicator has carried since gemma_hip_comm_init; the seconds are host wall time around synchronised collectives and stay
 * zero unless GEMMA_HIP_COMM_TIMING=1 is in the environment */
typedef struct {
  long allreduce_calls, allreduce_pieces, bcast_calls, bcast_pieces;
  double allreduce_bytes, bcast_bytes, allreduce_s, bcast_s;
} gemma_comm_stats;
int gemma_hip_comm_stats(gemma_comm_stats *out);

/* ---- measurement -------------------------------------------------------- */
enum { GEMMA_STAGE_INGEST = 0, GEMMA_STAGE_UTX_GEMM = 1, GEMMA_STAGE_ASSOC = 2,
       GEMMA_STAGE_KIN_GEMM = 3, GEMMA_STAGE_EIGH = 4,
       GEMMA_STAGE_UTX_POST = 5, /* int8-digit path only: int32 digit sums -> fp64 U^T x */
       GEMMA_STAGE_COUNT = 6 };
/* when on, every kernel of a stage is bracketed by hipEvents on its launch stream */
int gemma_hip_profile_enable(int on);
/* synchronises, then returns accumulated GPU milliseconds and launch count; reset != 0 clears */
int gemma_hip_profile_read(int stage, double *total_ms, long *launches, int reset);

/* ---- diagnostics: the stages of gemma_hip_eigh in isolation (host pointers; tests only) ---- */
/* Householder tridiagonalisation: d[n], e[n-1], tau[n], VT (n x n, row j = reflector u_j; may be NULL) */
int gemma_hip_dbg_tridiag(const double *G, size_t n, double *d, double *e, double *tau, double *VT);
/* divide and conquer on a symmetric tridiagonal: w[n] ascending, ZT (n x n, row k = eigenvector k) */
int gemma_hip_dbg_stedc(const double *d, const double *e, size_t n, double *w, double *ZT);
/* two-stage reduction of the eigensolver alone (n even, >= 384): band (n x 129, row j = B(j..j+128, j)) after the dense ->
 * band stage, d[n], e[n-1] after the bulge chase */
int gemma_hip_dbg_eigh2(const double *G, size_t n, double *band, double *d, double *e);
/* stage seconds of the last eigendecomposition that ran with GEMMA_HIP_EIGH_TIMING=1 in the environment: t8 = {reduction (to band /
 * to tridiagonal), bulge chase, divide & conquer, back-transformation Q2, Q1 (one-stage: all of it), sort + transpose, n, 1|2 stages} */
int gemma_hip_dbg_eigh_last(double *t8);
/* the U^T x stage of gemma_hip_lmm_batch alone (after lmm_setup; host pointers): UtX is l x n row-major, row s =
 * (U^T x_s)^T of the mean-imputed SNP s (the column fast_dgemm("T","N",U,Xlarge) produces, GEMMA src/lmm.cpp:1521).
 * path 0: fp64 MFMA GEMM; path 1: exact int8-digit product where the input allows it (PLINK 2-bit; fp64 rows of hard calls;
 * fp64 rows of fixed-point dosages k/100 or k/1000 -- csrc/i8gemm.hip.h), else the fp64 GEMM */
int gemma_hip_dbg_utx(int geno_kind, const void *geno, size_t l, size_t ld, int path, double *UtX);
/* what the last U^T x (lmm_batch*, dbg_utx) ran on: 0 fp64 MFMA GEMM (src/lmm.cpp:1521 as one fp64 product), 1 int8-digit
 * product of hard calls, 2 / 3 int8-digit product of fixed-point dosages k/100 / k/1000 (BIMBAM mean genotypes,
 * doc/manual.tex:398-404) */
int gemma_hip_dbg_last_utx_path(int *path);
/* the MATRIX KERNEL that product launched, as the library's own launch site recorded it (bench.py labels `roofline.kernel` from this
 * and refuses counter files taken on another kernel -- VERDICT r4 #1b).  `name` is the kernel's symbol as rocprofv3 prints it;
 * `launches` counts launches since the library was loaded. */
enum {
  GEMMA_UTX_KERNEL_DGEMM_F64 = 0,    /* dgemm_mfma_glds_kernel: src/lmm.cpp:1521 as one fp64 MFMA product */
  GEMMA_UTX_KERNEL_DENSE_I8 = 1,     /* i8gemm_packed_kernel_t<true>: digit products, mask product dense (GEMMA_HIP_I8_SPARSE=0) */
  GEMMA_UTX_KERNEL_SPARSE_BYTES = 2, /* i8gemm_sparse_kernel: mask product on the 2:4 sparse MFMA, byte genotypes (=1) */
  GEMMA_UTX_KERNEL_RECORDS_R32 = 3,  /* i8gemm_sparse2_kernel: records, 32-row matrix instructions (GEMMA_HIP_I8_ROWS=32) */
  GEMMA_UTX_KERNEL_RECORDS_R16 = 4,  /* i8gemm_sparse2_r16_kernel: records, v_mfma_i32_16x16x64_i8 + v_smfmac_i32_16x16x128_i8 (default) */
  GEMMA_UTX_KERNEL_DOSAGE_I8 = 5,    /* i8gemm_packed_kernel_t<false, true>: byte planes of fixed-point dosages, 32-row instructions (GEMMA_HIP_DOSAGE_ROWS=32) */
  GEMMA_UTX_KERNEL_DOSAGE_I8_R16 = 6, /* i8gemm_dense16_kernel_t<true>: the same planes on v_mfma_i32_16x16x64_i8 (default) */
  GEMMA_UTX_KERNEL_COUNT = 7
};
typedef struct {
  int variant;  /* GEMMA_UTX_KERNEL_* */
  int rows;     /* rows of the matrix instruction: 16 | 32 (0: fp64) */
  int digits;   /* base-256 digits of U in the product (0: fp64) */
  int fuse;     /* 1: two digits per int32 plane */
  int raster;   /* row blocks of the cross-XCD raster (0: per-XCD ranges) */
  long launches;
  char name[64];
} gemma_utx_kernel_info;
int gemma_hip_dbg_last_utx_kernel(gemma_utx_kernel_info *info);
/* Blocks WITHOUT a missing call (round 6).  With x = g + mean * m the mask product U^T (mean * m) is identically zero for them; the
 * records path notices on the device -- the kernel that builds the records sets a flag when it meets a missing call -- and the launch
 * site queues both forms of the 16-row kernel, of which the one that is not the block's returns at once: complete blocks take the
 * genotype product alone (i8gemm_sparse2_r16_g_kernel) and the digit combine reads no mask rows.  U^T x is the same, bit for bit (a
 * mask product of zeros adds +0.0); GEMMA_HIP_I8_COMPLETE=0 sends every block through both products.  *any: that flag for the last
 * records product of a plain batch (1 / 0; -1: form off or no such product yet).  Synchronises the device. */
int gemma_hip_dbg_last_block_missing(int *any);
/* re-read the GEMMA_HIP_* switches of the batch path.  The library reads them once per setup (gemma_hip_init, lmm_setup*, lm_setup,
 * mvlmm_null / mvlmm_set, kin_begin), never per launch; a caller that changes one between two batches of ONE setup calls this. */
int gemma_hip_reload_env(void);
/* base-256 digits of U the int8 product uses at this n: 7 (U to 2^-56 of the column maximum); 6 from n = 16384 up
 * (U rounded to 2^-48 of the column maximum -- see gemma_hip_lmm_batch); GEMMA_HIP_I8_DIGITS=6|7 and GEMMA_HIP_I8_FORM=7g6m override */
int gemma_hip_dbg_i8_digits(size_t n, int *digits);

#ifdef __cplusplus
}
#endif
#endif /* GEMMA_HIP_H */
