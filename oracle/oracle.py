"""Python side of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under gemma_amd/ does.  It drives the C restatement
(oracle/gemma_oracle.c) and supplies the pieces the reference delegates to OpenBLAS
through numpy/scipy's bundled OpenBLAS (cblas_dgemm == numpy matmul,
dsyevr_ == scipy.linalg.lapack.dsyevr), plus restatements of the reference's file
readers / QC filters so that the reference's own golden values (BXD) can be reproduced.
The whole restatement is pinned against the reference itself (oracle/_ref/gemma, built by
`make -C oracle ref` from /root/reference/src unchanged): tests/test_reference_pin.py.

File:line citations are relative to /root/reference.
"""
import ctypes as C
import gzip
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LMM_BATCH_SIZE = 20000  # src/lmm.h:33
K_BATCH_SIZE = 20000  # src/param.h:32


class SumStat(C.Structure):  # == SUMSTAT src/param.h:54-66
    _fields_ = [(k, C.c_double) for k in
                ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score", "logl_H1")]


SUMSTAT_DTYPE = np.dtype([(k, "f8") for k in
                          ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score",
                           "logl_H1")])


class MvCfg(C.Structure):
    """orc_mv_cfg: the MVLMM members CopyFromParam fills (src/mvlmm.cpp:51-90) with PARAM's defaults"""
    _fields_ = [("em_iter", C.c_size_t), ("nr_iter", C.c_size_t), ("n_region", C.c_size_t), ("em_prec", C.c_double),
                ("nr_prec", C.c_double), ("l_min", C.c_double), ("l_max", C.c_double), ("p_nr", C.c_double),
                ("crt", C.c_size_t)]


def mv_cfg(em_iter=10000, nr_iter=100, em_prec=1e-4, nr_prec=1e-4, l_min=1e-5, l_max=1e5, n_region=10, p_nr=1e-3, crt=0):
    return MvCfg(em_iter, nr_iter, n_region, em_prec, nr_prec, l_min, l_max, p_nr, crt)


def build():
    so = os.path.join(_HERE, "libgemma_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("gemma_oracle.c", "mvlmm_oracle.c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgemma_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.orc_GetabIndex.restype = C.c_size_t
        L.orc_GetabIndex.argtypes = [C.c_size_t] * 3
        L.orc_safe_sqrt.restype = C.c_double
        L.orc_safe_sqrt.argtypes = [C.c_double]
        L.orc_cdf_fdist_Q.restype = C.c_double
        L.orc_cdf_fdist_Q.argtypes = [C.c_double] * 3
        L.orc_cdf_chisq_Q1.restype = C.c_double
        L.orc_cdf_chisq_Q1.argtypes = [C.c_double]
        L.orc_CalcLambda_null.argtypes = [C.c_char, C.c_size_t, C.c_size_t, dp, dp, dp, C.c_double,
                                          C.c_double, C.c_size_t, dp, dp]
        L.orc_CalcPve.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, C.c_double, C.c_double, dp, dp]
        L.orc_CalcLmmVgVeBeta.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, C.c_double, dp, dp, dp, dp]
        L.orc_lmm_batch.argtypes = [C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_size_t,
                                    C.c_double, C.c_double, C.c_size_t, C.c_double, C.c_double, C.c_int,
                                    dp, C.POINTER(SumStat), C.POINTER(C.c_long)]
        L.orc_lm_batch.argtypes = [C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_size_t, C.POINTER(SumStat)]
        L.orc_newton_step_rel.restype = None
        L.orc_newton_step_rel.argtypes = [C.c_char, C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_size_t, dp, dp, dp]
        L.orc_gene_batch.argtypes = [C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_size_t, C.c_double, C.c_double,
                                     C.c_size_t, C.POINTER(SumStat)]
        L.orc_gene_batch.restype = None
        L.orc_gxe_batch.argtypes = [C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, dp, dp, C.POINTER(C.c_int), C.c_size_t,
                                    C.c_double, C.c_double, C.c_size_t, C.c_double, C.POINTER(SumStat)]
        L.orc_gxe_batch.restype = None
        L.orc_impute_mean.argtypes = [dp, C.c_size_t, C.c_size_t]
        L.orc_kin_prepare.argtypes = [dp, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_bed_decode.restype = C.c_size_t
        L.orc_bed_decode.argtypes = [C.POINTER(C.c_ubyte), C.c_size_t, C.POINTER(C.c_int), dp]
        L.orc_CenterMatrix.argtypes = [dp, C.c_size_t]
        L.orc_zero_small_eval.restype = C.c_double
        L.orc_zero_small_eval.argtypes = [dp, C.c_size_t]
        L.orc_dgemm.argtypes = [C.c_char, C.c_char, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, dp,
                                C.c_size_t, dp, C.c_size_t, C.c_double, dp, C.c_size_t]
        sz, cd = C.c_size_t, C.c_double
        L.orc_cdf_chisq_Q.restype = cd
        L.orc_cdf_chisq_Q.argtypes = [cd, cd]
        L.orc_mph_em.restype = cd
        L.orc_mph_em.argtypes = [C.c_char, sz, cd, sz, sz, sz, dp, dp, dp, dp, dp, dp]
        L.orc_mph_nr.restype = cd
        L.orc_mph_nr.argtypes = [C.c_char, sz, cd, sz, sz, sz, dp, dp, dp, dp, dp, dp]
        L.orc_mph_nr_crt.restype = cd
        L.orc_mph_nr_crt.argtypes = [C.c_char, sz, cd, sz, sz, sz, dp, dp, dp, dp, dp, dp, dp]
        L.orc_cdf_chisq_Qinv.restype = cd
        L.orc_cdf_chisq_Qinv.argtypes = [cd, cd]
        L.orc_pcrt.restype = cd
        L.orc_pcrt.argtypes = [C.c_int, sz, cd, cd, cd, cd]
        L.orc_mph_calcp.restype = cd
        L.orc_mph_calcp.argtypes = [sz, sz, sz, dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_mph_dev.restype = cd
        L.orc_mph_dev.argtypes = [C.c_char, sz, sz, sz, dp, dp, dp, dp, dp, dp, dp]
        L.orc_mvlmm_null.restype = None
        L.orc_mvlmm_null.argtypes = [C.POINTER(MvCfg), sz, sz, sz, dp, dp, dp] + [dp] * 8
        L.orc_mvlmm_batch_gxe.restype = None
        L.orc_mvlmm_batch_gxe.argtypes = [C.c_int, C.POINTER(MvCfg), sz, sz, sz, dp, dp, dp, dp, dp, sz, dp, dp, dp, dp]
        L.orc_mvlmm_batch.restype = None
        L.orc_mvlmm_batch.argtypes = [C.c_int, C.POINTER(MvCfg), sz, sz, sz, dp, dp, dp, dp, sz, dp, dp, dp, cd, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------- scalars
def GetabIndex(a, b, n_cvt):
    return lib().orc_GetabIndex(a, b, n_cvt)


def fdist_Q(x, nu1, nu2):
    return lib().orc_cdf_fdist_Q(x, nu1, nu2)


def chisq_Q1(x):
    return lib().orc_cdf_chisq_Q1(x)


# --------------------------------------------------------------------------- dense pieces
def dgemm(ta, tb, alpha, A, B, beta, Cm):
    """Contract of fast_cblas_dgemm (src/fastblas.cpp:66-170), plain loops (KAT sizes only)."""
    A = _c64(A); B = _c64(B); Cm = _c64(Cm)
    M, N = Cm.shape
    K = A.shape[0] if ta in "Tt" else A.shape[1]
    lib().orc_dgemm(ta.encode(), tb.encode(), M, N, K, alpha, _dp(A), A.shape[1], _dp(B), B.shape[1],
                    beta, _dp(Cm), Cm.shape[1])
    return Cm


def center_matrix(G):
    """CenterMatrix, src/mathfunc.cpp:147-177."""
    G = _c64(G).copy()
    lib().orc_CenterMatrix(_dp(G), G.shape[0])
    return G


def eigen_decomp_zeroed(G):
    """EigenDecomp_Zeroed, src/lapack.cpp:260-291 -> dsyevr_('V','A','L',abstol=1e-7) :205-222 and the
    transpose at :228, so that eigenvector k is column k of row-major U. Returns (U, eval, trace_G)."""
    from scipy.linalg import lapack
    G = _c64(G)
    # the reference passes the row-major buffer as column-major with UPLO='L': that reads the
    # upper triangle of the row-major matrix.  G is symmetric, so pass G.T (a column-major view).
    w, z, m, isuppz, info = lapack.dsyevr(G.T, compute_v=1, range="A", lower=1, abstol=1.0e-7,
                                          overwrite_a=0)
    assert info == 0, "dsyevr failed"
    # `z` holds eigenvectors in its columns (Fortran order); the reference's evec->data is that same
    # buffer read row-major (= z.T) and then transposed => U == z with eigenvector k in column k.
    U = np.ascontiguousarray(z)
    ev = np.ascontiguousarray(w, dtype=np.float64)
    trace_G = lib().orc_zero_small_eval(_dp(ev), ev.size)
    return U, ev, trace_G


def calc_UtX(U, X):
    """CalcUtX, src/mathfunc.cpp:504-506: UtX = U^T X via cblas_dgemm (numpy matmul -> OpenBLAS)."""
    return np.ascontiguousarray(U.T @ X)


# --------------------------------------------------------------------------- null model
def calc_lambda_null(func, ev, UtW, Uty, l_min=1e-5, l_max=1e5, n_region=10):
    ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty)
    n, c = UtW.shape
    lam = C.c_double(); logl = C.c_double()
    lib().orc_CalcLambda_null(func.encode(), n, c, _dp(ev), _dp(UtW), _dp(Uty), l_min, l_max, n_region,
                              C.byref(lam), C.byref(logl))
    return lam.value, logl.value


def calc_pve(ev, UtW, Uty, lam, trace_G):
    ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty)
    n, c = UtW.shape
    a = C.c_double(); b = C.c_double()
    lib().orc_CalcPve(n, c, _dp(ev), _dp(UtW), _dp(Uty), lam, trace_G, C.byref(a), C.byref(b))
    return a.value, b.value


def calc_vg_ve_beta(ev, UtW, Uty, lam):
    ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty)
    n, c = UtW.shape
    vg = C.c_double(); ve = C.c_double()
    beta = np.zeros(c); se = np.zeros(c)
    lib().orc_CalcLmmVgVeBeta(n, c, _dp(ev), _dp(UtW), _dp(Uty), lam, C.byref(vg), C.byref(ve), _dp(beta),
                              _dp(se))
    return vg.value, ve.value, beta, se


# --------------------------------------------------------------------------- association
def lmm_batch_UtX(a_mode, ev, UtW, Uty, UtX_snpmajor, l_mle_null=0.0, logl_mle_H0=0.0, l_min=1e-5,
                  l_max=1e5, n_region=10, plink_nan_rule=0, carry=None, want_diag=False):
    """batch_compute's per-SNP loop, src/lmm.cpp:1526-1562, on a precomputed SNP-major UtX (l x n)."""
    ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty); UtX = _c64(UtX_snpmajor)
    n, c = UtW.shape
    l = UtX.shape[0]
    out = np.zeros(l, dtype=SUMSTAT_DTYPE)
    diag = np.zeros((l, 3), dtype=np.int64) if want_diag else None
    cr = np.zeros(2) if carry is None else carry
    lib().orc_lmm_batch(a_mode, n, c, _dp(ev), _dp(UtW), _dp(Uty), _dp(UtX), l, l_min, l_max, n_region,
                        l_mle_null, logl_mle_H0, plink_nan_rule, _dp(cr),
                        out.ctypes.data_as(C.POINTER(SumStat)),
                        diag.ctypes.data_as(C.POINTER(C.c_long)) if want_diag else None)
    return (out, diag) if want_diag else out


def newton_step_rel(func, ev, UtW, Uty, UtX_snpmajor, lambdas):
    """(relative Newton step of CalcLambda's polish from lambdas[s], logf(lambdas[s])) per SNP row -- test aid for the
    two-tier lambda criterion (orc_newton_step_rel)."""
    ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty); UtX = _c64(UtX_snpmajor); lam = _c64(lambdas)
    n, c = UtW.shape
    l = UtX.shape[0]
    step = np.zeros(l); logf = np.zeros(l)
    lib().orc_newton_step_rel(func.encode(), n, c, _dp(ev), _dp(UtW), _dp(Uty), _dp(UtX), l, _dp(lam), _dp(step), _dp(logf))
    return step, logf


def gene_analyze(a_mode, U, ev, UtW, Utx, Y, l_min=1e-5, l_max=1e5, n_region=10):
    """LMM::AnalyzeGene, src/lmm.cpp:1365-1471: rows of Y (genes x n) are phenotypes, Utx the fixed tested variable."""
    ev = _c64(ev); UtW = _c64(UtW); Utx = _c64(Utx)
    n, c = UtW.shape
    UtY = np.ascontiguousarray(_c64(Y) @ _c64(U))  # row g = (U^T y_g)^T, gsl_blas_dgemv(CblasTrans, U, y) at :1415
    l = UtY.shape[0]
    out = np.zeros(l, dtype=SUMSTAT_DTYPE)
    lib().orc_gene_batch(a_mode, n, c, _dp(ev), _dp(UtW), _dp(Utx), _dp(UtY), l, l_min, l_max, n_region,
                         out.ctypes.data_as(C.POINTER(SumStat)))
    return out


def gxe_analyze(a_mode, U, ev, UtW, Uty, env, X_snpmajor_nan, l_mle_null=0.0, l_min=1e-5, l_max=1e5, n_region=10):
    """LMM::AnalyzeBimbamGXE / AnalyzePlinkGXE, src/lmm.cpp:2283-2608: covariates [W, env, x_s], tested x_s . env.
    Feeder part here (:2316-2366): mean imputation, recode 2 - x when x_mean > 1, the two rotations."""
    U = _c64(U); ev = _c64(ev); UtW = _c64(UtW); Uty = _c64(Uty); env = _c64(env)
    n, c = UtW.shape
    X = impute_mean(X_snpmajor_nan)
    with np.errstate(invalid="ignore"):
        xm = np.nanmean(_c64(X_snpmajor_nan), axis=1)
    flip = np.ascontiguousarray((xm > 1).astype(np.int32))
    X = np.where(flip[:, None] == 1, 2.0 - X, X)
    UtX = np.ascontiguousarray(X @ U)
    UtZ = np.ascontiguousarray((X * env[None, :]) @ U)
    UtWe = np.ascontiguousarray(np.hstack([UtW, (U.T @ env)[:, None]]))
    l = X.shape[0]
    out = np.zeros(l, dtype=SUMSTAT_DTYPE)
    lib().orc_gxe_batch(a_mode, n, c, _dp(ev), _dp(UtWe), _dp(Uty), _dp(UtX), _dp(UtZ),
                        flip.ctypes.data_as(C.POINTER(C.c_int)), l, l_min, l_max, n_region, l_mle_null,
                        out.ctypes.data_as(C.POINTER(SumStat)))
    return out


def lm_analyze(a_mode, W, y, X_snpmajor_nan):
    """LM::AnalyzeBimbam / AnalyzePlink (src/lm.cpp:382-640): ordinary regression per SNP, a_mode 51..54."""
    W = _c64(W).reshape(len(y), -1); y = _c64(y)
    n, c = W.shape
    WtWi = np.ascontiguousarray(np.linalg.inv(W.T @ W))
    X = impute_mean(X_snpmajor_nan)
    out = np.zeros(X.shape[0], dtype=SUMSTAT_DTYPE)
    lib().orc_lm_batch(a_mode, n, c, _dp(W), _dp(WtWi), _dp(y), _dp(X), X.shape[0], out.ctypes.data_as(C.POINTER(SumStat)))
    return out


def impute_mean(X_snpmajor):
    X = _c64(X_snpmajor).copy()
    lib().orc_impute_mean(_dp(X), X.shape[0], X.shape[1])
    return X


def lmm_analyze(a_mode, U, ev, UtW, Uty, X_snpmajor_nan, batch=LMM_BATCH_SIZE, **kw):
    """LMM::Analyze, src/lmm.cpp:1474-1658: mean-impute (:1590-1618), batches of 20000,
    UtX = U^T X by cblas_dgemm (:1521), then the serial per-SNP loop.  X is SNP-major (p x n)
    over the analysed individuals with NaN = missing."""
    p = X_snpmajor_nan.shape[0]
    outs = []
    carry = np.zeros(2)
    for s0 in range(0, p, batch):
        Xb = impute_mean(X_snpmajor_nan[s0:s0 + batch])
        UtX = np.ascontiguousarray(Xb @ U)  # (l x n): row s = (U^T x_s)^T
        outs.append(lmm_batch_UtX(a_mode, ev, UtW, Uty, UtX, carry=carry, **kw))
    return np.concatenate(outs) if outs else np.zeros(0, dtype=SUMSTAT_DTYPE)


# --------------------------------------------------------------------------- kinship
def kin_prepare(X_snpmajor_nan, k_mode):
    X = _c64(X_snpmajor_nan).copy()
    lib().orc_kin_prepare(_dp(X), X.shape[0], X.shape[1], k_mode)
    return X


def calc_kin(X_snpmajor_nan, k_mode=1, batch=K_BATCH_SIZE):
    """BimbamKin / PlinkKin, src/gemma_io.cpp:1418-1597 / :1599-1738: per 20000 SNPs
    K += Xb Xb^T (cblas_dgemm, :1554), finally K /= ns_test (:1570)."""
    p, n = X_snpmajor_nan.shape
    K = np.zeros((n, n))
    for s0 in range(0, p, batch):
        Xb = kin_prepare(X_snpmajor_nan[s0:s0 + batch], k_mode)
        K += Xb.T @ Xb
    K *= 1.0 / float(p)
    return K


def round10(M):
    """PARAM::WriteMatrix precision(10) (src/param.cpp:1899) -> ReadFile_kin atof round trip."""
    flat = np.array([float("%.10g" % v) for v in np.asarray(M).ravel()])
    return flat.reshape(np.asarray(M).shape)


# --------------------------------------------------------------------------- readers + QC
def _tok(line):
    return line.replace(",", " ").replace("\t", " ").split()


def read_pheno(path, col=1):
    """ReadFile_pheno src/gemma_io.cpp:386-444 (one column). Returns (values, indicator)."""
    vals, ind = [], []
    with open(path) as f:
        for line in f:
            t = _tok(line)
            if not t:
                continue
            if t[col - 1] == "NA":
                vals.append(-9.0); ind.append(0)
            else:
                vals.append(float(t[col - 1])); ind.append(1)
    return np.array(vals), np.array(ind, dtype=np.int32)


def read_cvt(path):
    """ReadFile_cvt src/gemma_io.cpp:446-511."""
    rows, ind = [], []
    with open(path) as f:
        for line in f:
            t = _tok(line)
            if not t:
                continue
            ind.append(0 if "NA" in t else 1)
            rows.append([(-9.0 if x == "NA" else float(x)) for x in t])
    return np.array(rows), np.array(ind, dtype=np.int32)


def process_cvt_phen(ind_pheno, cvt=None, ind_cvt=None):
    """PARAM::ProcessCvtPhen + CheckCvt, src/param.cpp:1993-2098 / :1937-1990:
    indicator_idv and the covariate matrix W (intercept column appended when no constant
    column exists)."""
    ind = ind_pheno.copy()
    if cvt is not None:
        ind = ind * ind_cvt
    sel = ind == 1
    if cvt is None:
        W = np.ones((int(sel.sum()), 1))
        return ind, W
    W = cvt[sel]
    const_cols = [j for j in range(W.shape[1]) if W[:, j].min() == W[:, j].max()]
    if len(const_cols) == W.shape[1]:
        W = np.ones((int(sel.sum()), 1))
    elif not const_cols:
        W = np.hstack([W, np.ones((W.shape[0], 1))])
    return ind, W


def read_bimbam_geno(path):
    """Mean-genotype file (src/gemma_io.cpp:706-793): id, a1, a0, g_1..g_n; NA = missing (NaN)."""
    op = gzip.open if path.endswith(".gz") else open
    rs, rows = [], []
    with op(path, "rt") as f:
        for line in f:
            t = _tok(line)
            if not t:
                continue
            rs.append(t[0])
            rows.append([float("nan") if x == "NA" else float(x) for x in t[3:]])
    return rs, np.array(rows, dtype=np.float64)


def calc_hwe(n_hom1, n_hom2, n_ab):
    """CalcHWE, src/mathfunc.cpp:546-627 (Wigginton et al. 2005 exact test)."""
    if n_hom1 + n_hom2 + n_ab == 0:
        return 1.0
    n_aa, n_bb = min(n_hom1, n_hom2), max(n_hom1, n_hom2)
    rare = 2 * n_aa + n_ab
    geno = n_ab + n_bb + n_aa
    het = np.zeros(rare + 1)
    mid = (rare * (2 * geno - rare)) // (2 * geno)
    if (rare & 1) ^ (mid & 1):
        mid += 1
    homr, homc = (rare - mid) // 2, geno - mid - (rare - mid) // 2
    het[mid] = 1.0
    tot = 1.0
    h = mid
    while h > 1:
        het[h - 2] = het[h] * h * (h - 1.0) / (4.0 * (homr + 1.0) * (homc + 1.0))
        tot += het[h - 2]
        homr += 1; homc += 1; h -= 2
    homr, homc = (rare - mid) // 2, geno - mid - (rare - mid) // 2
    h = mid
    while h <= rare - 2:
        het[h + 2] = het[h] * 4.0 * homr * homc / ((h + 2.0) * (h + 1.0))
        tot += het[h + 2]
        homr -= 1; homc -= 1; h += 2
    het /= tot
    p = het[het <= het[n_ab]].sum()
    return min(p, 1.0)


def qc_snps(G_all, indicator_idv, W, maf_level=0.01, miss_level=0.05, r2_level=0.9999):
    """First-pass SNP filters of ReadFile_geno, src/gemma_io.cpp:753-853 (hwe off by default),
    statistics over analysed individuals. G_all: p x ni_total with NaN. Returns (indicator_snp, maf,
    n_miss)."""
    sel = indicator_idv == 1
    G = G_all[:, sel]
    p, ni_test = G.shape
    WtWi = np.linalg.inv(W.T @ W)
    ind = np.zeros(p, dtype=np.int32)
    mafs = np.zeros(p); nmiss = np.zeros(p, dtype=np.int64)
    for t in range(p):
        g = G[t]
        miss = np.isnan(g)
        n_miss = int(miss.sum())
        obs = g[~miss]
        maf = obs.sum() / (2.0 * (ni_test - n_miss)) if ni_test > n_miss else float("nan")
        mafs[t] = maf; nmiss[t] = n_miss
        if n_miss / ni_test > miss_level:
            continue
        if (maf < maf_level or maf > 1.0 - maf_level) and maf_level != -1:
            continue
        if obs.size == 0 or np.all(obs == obs[0]):  # flag_poly != 1, :818
            continue
        x = np.where(miss, maf * 2.0, g)
        Wtx = W.T @ x
        v_x = x @ x
        v_w = Wtx @ (WtWi @ Wtx)
        if W.shape[1] != 1 and v_w / v_x > r2_level:
            continue
        ind[t] = 1
    return ind, mafs, nmiss


def read_bed(prefix):
    """PLINK .bed/.bim/.fam (src/gemma_io.cpp:514-635, :918-997): returns (raw SNP-major byte matrix
    p x ceil(n/4), ni_total, phenotype column 6 with NA/-9 -> indicator)."""
    fam = [l.split() for l in open(prefix + ".fam") if l.strip()]
    ni_total = len(fam)
    ph, ind = [], []
    for r in fam:
        if r[5] == "NA" or r[5] == "-9":
            ph.append(-9.0); ind.append(0)
        else:
            ph.append(float(r[5])); ind.append(1)
    nsnp = sum(1 for l in open(prefix + ".bim") if l.strip())
    n_bit = (ni_total + 3) // 4
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)[3:]
    raw = raw[: nsnp * n_bit].reshape(nsnp, n_bit)
    return raw, ni_total, np.array(ph), np.array(ind, dtype=np.int32)


def bed_decode(raw_rows, ni_total, indicator=None):
    """Decode SNP-major .bed rows to doubles (NaN = missing) through the C restatement."""
    raw_rows = np.ascontiguousarray(raw_rows, dtype=np.uint8)
    p = raw_rows.shape[0]
    n_out = ni_total if indicator is None else int((indicator != 0).sum())
    out = np.empty((p, n_out))
    indp = None if indicator is None else np.ascontiguousarray(indicator, dtype=np.int32).ctypes.data_as(
        C.POINTER(C.c_int))
    for s in range(p):
        lib().orc_bed_decode(raw_rows[s].ctypes.data_as(C.POINTER(C.c_ubyte)), ni_total, indp,
                             _dp(out[s]))
    return out


def qc_snps_bed(G_test, W, maf_level=0.01, miss_level=0.05, r2_level=0.9999, hwe_level=0.0):
    """ReadFile_bed filters, src/gemma_io.cpp:942-1049: G_test is p x ni_test with NaN."""
    p, ni_test = G_test.shape
    WtWi = np.linalg.inv(W.T @ W)
    ind = np.zeros(p, dtype=np.int32)
    for t in range(p):
        g = G_test[t]
        miss = np.isnan(g)
        n_miss = int(miss.sum())
        obs = g[~miss]
        maf = obs.sum() / (2.0 * (ni_test - n_miss))
        if n_miss / ni_test > miss_level:
            continue
        if (maf < maf_level or maf > 1.0 - maf_level) and maf_level != -1:
            continue
        n0 = int((obs == 0).sum()); n1 = int((obs == 1).sum()); n2 = int((obs == 2).sum())
        if (n0 + n1) == 0 or (n1 + n2) == 0 or (n2 + n0) == 0:  # :1017-1020
            continue
        if hwe_level != 0 and maf_level != -1 and calc_hwe(n0, n2, n1) < hwe_level:  # :1022-1027
            continue
        x = np.where(miss, maf * 2.0, g)
        Wtx = W.T @ x
        v_w = Wtx @ (WtWi @ Wtx)
        if W.shape[1] != 1 and v_w / (x @ x) > r2_level:
            continue
        ind[t] = 1
    return ind


# --------------------------------------------------------------------------- end-to-end
def run_lmm(a_mode, G_all, indicator_idv, indicator_snp, y_all, W, K_full, maf_note=None):
    """BatchRun's LMM branch, src/gemma.cpp:2557-2830, from in-memory inputs.
    K_full: ni_total x ni_total (already through round10 when emulating the cXX.txt hand-off)."""
    sel = indicator_idv == 1
    y = y_all[sel]
    G = K_full[np.ix_(sel, sel)]  # ReadFile_kin sub-selection, src/gemma_io.cpp:1205-1243
    G = center_matrix(G)
    U, ev, trace_G = eigen_decomp_zeroed(G)
    UtW = calc_UtX(U, W)
    Uty = calc_UtX(U, y.reshape(-1, 1)).ravel()
    l_mle_null, logl_mle_H0 = calc_lambda_null("L", ev, UtW, Uty)
    l_remle_null, logl_remle_H0 = calc_lambda_null("R", ev, UtW, Uty)
    pve, pve_se = calc_pve(ev, UtW, Uty, l_remle_null, trace_G)
    X = G_all[indicator_snp == 1][:, sel]
    stats = lmm_analyze(a_mode, U, ev, UtW, Uty, X, l_mle_null=l_mle_null, logl_mle_H0=logl_mle_H0)
    null = dict(l_mle_null=l_mle_null, logl_mle_H0=logl_mle_H0, l_remle_null=l_remle_null,
                logl_remle_H0=logl_remle_H0, pve=pve, pve_se=pve_se, trace_G=trace_G)
    return stats, null, dict(U=U, eval=ev, UtW=UtW, Uty=Uty, X=X)


# --------------------------------------------------------------------------- the reference itself (oracle/_ref)
_REF = None


def ref_lib():
    """oracle/_ref/libgemma_ref.so -- the reference's own objects (built by `make -C oracle ref` from /root/reference/src,
    unchanged) plus the C entry points of oracle/ref_bridge.cpp -- or None when it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libgemma_ref.so")
        if not os.path.exists(path):
            return None
        try:
            _REF = C.CDLL(path)
        except OSError:
            return None
        P = C.POINTER(C.c_double)
        _REF.ref_lmm_analyze.restype = C.c_long
        _REF.ref_lmm_analyze.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t] + [P] * 7 + [C.c_double, C.c_double, C.c_size_t,
                                                                                               C.c_double, C.c_double, P]
    return _REF


def ref_eigen_decomp_zeroed(G):
    """The reference's own EigenDecomp_Zeroed (src/lapack.cpp:260-291, dsyevr_ of the OpenBLAS the reference binary is linked
    against) -> (U, eval, trace_G); G is copied."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libgemma_ref.so not built")
    G = _c64(G).copy()
    n = G.shape[0]
    U, ev = np.empty((n, n)), np.empty(n)
    R.ref_eigen_decomp_zeroed.restype = C.c_double
    R.ref_eigen_decomp_zeroed.argtypes = [C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    tr = R.ref_eigen_decomp_zeroed(n, _dp(G), _dp(U), _dp(ev))
    return U, ev, tr


def ref_plink_kin(bed_path, ni, ns, k_mode=1):
    """The reference's own PlinkKin (src/gemma_io.cpp:1599-1738) on a .bed file (3 magic bytes + ns rows of ceil(ni/4) bytes)."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libgemma_ref.so not built")
    K = np.zeros((ni, ni))
    R.ref_plink_kin.restype = C.c_int
    R.ref_plink_kin.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    rc = R.ref_plink_kin(os.fsencode(bed_path), ni, ns, k_mode, _dp(K))
    if rc:
        raise RuntimeError("PlinkKin failed on " + bed_path)
    return K


def ref_blas_threads():
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    return int(C.CDLL(libs[0]).scipy_openblas_get_num_threads()) if libs else 0


def ref_lmm_analyze(a_mode, U, ev, UtW, Uty, X_snpmajor_nan, l_mle_null=0.0, logl_mle_H0=0.0, l_min=1e-5, l_max=1e5, n_region=10):
    """LMM::Analyze of the reference (src/lmm.cpp:1474-1658), called in-process on SNP-major rows over the analysed
    individuals (NaN = NA).  Returns SUMSTAT records like lmm_analyze."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libgemma_ref.so not built")
    U, ev, Uty = _c64(U), _c64(ev), _c64(Uty)
    n = U.shape[0]
    UtW = _c64(np.asarray(UtW).reshape(n, -1))
    X = _c64(X_snpmajor_nan)
    W, y = _c64(U @ UtW), _c64(U @ Uty)  # only echoed by the reference's debug writer
    out = np.zeros((X.shape[0], 8))
    got = R.ref_lmm_analyze(a_mode, n, UtW.shape[1], X.shape[0], _dp(U), _dp(ev), _dp(UtW), _dp(Uty), _dp(W), _dp(y), _dp(X),
                            l_min, l_max, n_region, l_mle_null, logl_mle_H0, _dp(out))
    assert got == X.shape[0], (got, X.shape[0])
    st = np.zeros(X.shape[0], dtype=[(k, "<f8") for k in ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score",
                                                        "logl_H1")])
    for j, k in enumerate(st.dtype.names):
        st[k] = out[:, j]
    return st


# --------------------------------------------------------------------------- multivariate LMM (mvlmm_oracle.c)
def chisq_Q(x, nu):
    return lib().orc_cdf_chisq_Q(float(x), float(nu))


def mv_use_lapack_basis(on=True):
    """EigenProc of the multivariate oracle through LAPACK's dsyevr_ (the OpenBLAS inside scipy, the routine oracle/_ref/gemma
    is linked against) instead of the Jacobi convention: the reference's ML EM depends on the eigenvector signs dsyevr
    returns (mvlmm_oracle.c, mv_eig), so only this mode reproduces its `-lmm 2/4` output on every SNP for d >= 3."""
    L = lib()
    L.orc_mv_set_lapack_dsyevr.argtypes = [C.c_void_p]
    L.orc_mv_set_lapack_dsyevr.restype = None
    if not on:
        L.orc_mv_set_lapack_dsyevr(None)
        return
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if not libs:
        raise RuntimeError("scipy's OpenBLAS not found")
    ob = C.CDLL(libs[0])
    L.orc_mv_set_lapack_dsyevr(C.cast(ob.scipy_dsyevr_, C.c_void_p))


def mph_em(func, max_iter, max_prec, ev, X, Y, Vg, Ve, B):
    """MphEM on X (c x n), Y (d x n); Vg, Ve, B (d x c) are updated in place; returns logl."""
    d, n = Y.shape
    return lib().orc_mph_em(func.encode(), max_iter, max_prec, n, d, X.shape[0], _dp(ev), _dp(X), _dp(Y), _dp(Vg), _dp(Ve),
                            _dp(B))


def mph_initial(cfg, ev, X, Y):
    """MphInitial (src/mvlmm.cpp:2763-2948): the starting point of the null block -> (Vg, Ve, B)"""
    d, n = Y.shape
    c = X.shape[0]
    Vg, Ve, B = np.zeros((d, d)), np.zeros((d, d)), np.zeros((d, c))
    L = lib()
    sz, dp, cd = C.c_size_t, C.POINTER(C.c_double), C.c_double
    L.orc_mph_initial.restype = None
    L.orc_mph_initial.argtypes = [sz, cd, sz, cd, sz, sz, sz, dp, dp, dp, cd, cd, sz, dp, dp, dp]
    L.orc_mph_initial(cfg.em_iter, cfg.em_prec, cfg.nr_iter, cfg.nr_prec, n, d, c, _dp(ev), _dp(_c64(X)), _dp(_c64(Y)), cfg.l_min,
                      cfg.l_max, cfg.n_region, _dp(Vg), _dp(Ve), _dp(B))
    return Vg, Ve, B


def mph_nr(func, max_iter, max_prec, ev, X, Y, Vg, Ve):
    d, n = Y.shape
    Hi = np.zeros((d * (d + 1), d * (d + 1)))
    ll = lib().orc_mph_nr(func.encode(), max_iter, max_prec, n, d, X.shape[0], _dp(ev), _dp(X), _dp(Y), _dp(Vg), _dp(Ve),
                          _dp(Hi))
    return ll, Hi


def mph_nr_crt(func, max_iter, max_prec, ev, X, Y, Vg, Ve):
    """MphNR with the Edgeworth correction factors (crt_a, crt_b, crt_c) of its last CalcDev call (src/mvlmm.cpp:2054-2331)"""
    d, n = Y.shape
    Hi = np.zeros((d * (d + 1), d * (d + 1)))
    crt = np.zeros(3)
    ll = lib().orc_mph_nr_crt(func.encode(), max_iter, max_prec, n, d, X.shape[0], _dp(ev), _dp(X), _dp(Y), _dp(Vg), _dp(Ve),
                              _dp(Hi), _dp(crt))
    return ll, Hi, crt


def chisq_Qinv(q, nu):
    return lib().orc_cdf_chisq_Qinv(float(q), float(nu))


def pcrt(mode, d, p, crt):
    """PCRT (src/mvlmm.cpp:2952-2970): mode 1 Wald, 2 LRT, 3 score"""
    return lib().orc_pcrt(int(mode), int(d), float(p), float(crt[0]), float(crt[1]), float(crt[2]))


def mph_calcp(ev, x, W, Y, Vg, Ve):
    d, n = Y.shape
    beta, Vbeta = np.zeros(d), np.zeros((d, d))
    p = lib().orc_mph_calcp(n, d, W.shape[0], _dp(ev), _dp(x), _dp(W), _dp(Y), _dp(Vg), _dp(Ve), _dp(beta), _dp(Vbeta))
    return p, beta, Vbeta


def mph_dev(func, ev, X, Y, Vg, Ve, want_dev=True):
    d, n = Y.shape
    g, H = np.zeros(d * (d + 1)), np.zeros((d * (d + 1), d * (d + 1)))
    null = C.POINTER(C.c_double)()
    ll = lib().orc_mph_dev(func.encode(), n, d, X.shape[0], _dp(ev), _dp(X), _dp(Y), _dp(Vg), _dp(Ve),
                           _dp(g) if want_dev else null, _dp(H) if want_dev else null)
    return ll, g, H


def mvlmm_null(cfg, ev, W, Y):
    """The null block of MVLMM::AnalyzeBimbam: dict with the REMLE and MLE fits (W: cw x n, Y: d x n)."""
    d, n = Y.shape
    cw = W.shape[0]
    o = {k: np.zeros((d, d)) for k in ("Vg_remle", "Ve_remle", "Vg_mle", "Ve_mle")}
    o["B_remle"], o["B_mle"] = np.zeros((d, cw)), np.zeros((d, cw))
    lr, lm = C.c_double(), C.c_double()
    lib().orc_mvlmm_null(C.byref(cfg), n, d, cw, _dp(ev), _dp(W), _dp(Y), _dp(o["Vg_remle"]), _dp(o["Ve_remle"]),
                         _dp(o["B_remle"]), C.cast(C.byref(lr), C.POINTER(C.c_double)), _dp(o["Vg_mle"]),
                         _dp(o["Ve_mle"]), _dp(o["B_mle"]), C.cast(C.byref(lm), C.POINTER(C.c_double)))
    o["logl_remle"], o["logl_mle"] = lr.value, lm.value
    return o


def mvlmm_batch(a_mode, cfg, ev, W, Y, UtX_snpmajor, null):
    """Per-SNP block; returns dict of arrays: beta (l x d), Vbeta/Vg/Ve (l x v), p_wald, p_lrt, p_score."""
    d, n = Y.shape
    l = UtX_snpmajor.shape[0]
    v = d * (d + 1) // 2
    out = np.zeros((l, 3 * v + d + 3))
    X = _c64(UtX_snpmajor)
    lib().orc_mvlmm_batch(a_mode, C.byref(cfg), n, d, W.shape[0], _dp(ev), _dp(W), _dp(Y), _dp(X), l, _dp(null["Vg_mle"]),
                          _dp(null["Ve_mle"]), _dp(null["B_mle"]), null["logl_mle"], _dp(out))
    return {"beta": out[:, :d], "Vbeta": out[:, d:d + v], "Vg": out[:, d + v:d + 2 * v], "Ve": out[:, d + 2 * v:d + 3 * v],
            "p_wald": out[:, d + 3 * v], "p_lrt": out[:, d + 3 * v + 1], "p_score": out[:, d + 3 * v + 2]}


def mvlmm_batch_gxe(a_mode, cfg, ev, W_env, Y, UtX_snpmajor, UtX2_snpmajor, null):
    """Per-SNP block of MVLMM::AnalyzeBimbamGXE (src/mvlmm.cpp:4253-4348): W_env = the covariate rows with U^T env last, null = the
    fit of (W, env); UtX2 = the rotated x o env rows."""
    d, n = Y.shape
    l = UtX_snpmajor.shape[0]
    v = d * (d + 1) // 2
    out = np.zeros((l, 3 * v + d + 3))
    X, X2 = _c64(UtX_snpmajor), _c64(UtX2_snpmajor)
    lib().orc_mvlmm_batch_gxe(a_mode, C.byref(cfg), n, d, W_env.shape[0], _dp(ev), _dp(W_env), _dp(Y), _dp(X), _dp(X2), l,
                              _dp(null["Vg_mle"]), _dp(null["Ve_mle"]), _dp(null["B_mle"]), _dp(out))
    return {"beta": out[:, :d], "Vbeta": out[:, d:d + v], "Vg": out[:, d + v:d + 2 * v], "Ve": out[:, d + 2 * v:d + 3 * v],
            "p_wald": out[:, d + 3 * v], "p_lrt": out[:, d + 3 * v + 1], "p_score": out[:, d + 3 * v + 2]}
