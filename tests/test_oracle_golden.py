"""The oracle against the reference's own known answers (CPU only)."""
import os

import numpy as np
import pytest

REF = "/root/reference"


def test_getabindex_kat(oracle):
    # test/src/unittests-math.cpp:17-25
    assert oracle.GetabIndex(1, 1, 3) == 0
    assert oracle.GetabIndex(1, 2, 3) == 1
    assert oracle.GetabIndex(1, 3, 3) == 2
    assert oracle.GetabIndex(2, 2, 3) == 5
    assert oracle.GetabIndex(2, 3, 3) == 6
    assert oracle.GetabIndex(3, 3, 3) == 9
    assert oracle.GetabIndex(2, 1, 3) == 1


def test_dgemm_kat(oracle):
    # test/src/unittests-math.cpp:74-120 (2000 x 200 x 1000, A[i]=i+1, B[i]=-i-1)
    m, k, n = 2000, 200, 1000
    A = (np.arange(m * k, dtype=np.float64) + 1).reshape(m, k)
    B = (-np.arange(k * n, dtype=np.float64) - 1).reshape(k, n)
    Cm = oracle.dgemm("N", "N", 1.0, A, B, 0.0, np.zeros((m, n)))
    assert np.trunc(Cm.flat[0]) == -2666620100.0
    assert np.trunc(Cm.flat[1]) == -2666640200.0
    assert np.trunc(Cm.flat[2003]) == -10627000400.0


def test_safe_sqrt(oracle):
    # src/mathfunc.cpp:122-131: |d| for d < 0.001 (incl. negatives)
    L = oracle.lib()
    assert L.orc_safe_sqrt(4.0) == 2.0
    assert L.orc_safe_sqrt(-1e-4) == pytest.approx(1e-2)
    assert L.orc_safe_sqrt(-4.0) == 2.0  # the reference's `fabs(d < 0.001)` quirk


def test_cdfs_against_scipy(oracle):
    from scipy import special, stats
    # F(1,df) upper tail = regularised incomplete beta (scipy.stats.f.sf is only ~1e-9 accurate)
    for df in (63, 995, 19998, 49998, 250001):
        for x in (1e-6, 0.3, 1.0, 5.0, 30.0, 200.0, 1500.0):
            exact = (1.0 - special.betainc(0.5, df / 2.0, x / (df + x))) if x < df else \
                special.betainc(df / 2.0, 0.5, df / (df + x))
            assert oracle.fdist_Q(x, 1.0, df) == pytest.approx(exact, rel=2e-11)
    for x in (-1.0, 0.0, 1e-8, 0.5, 1.0, 3.84, 50.0, 400.0):
        assert oracle.chisq_Q1(x) == pytest.approx(stats.chi2.sf(x, 1) if x > 0 else 1.0, rel=1e-12)


def test_bxd_fixture_reproduces_reference_goldens(bxd):
    # test/dev_tests.rb:42-43,53-54 (reference asserts with 1e-3 abs; the oracle hits every digit)
    st2, st9 = bxd["stat_mode2"], bxd["stat_mode9"]
    assert "%.6e" % st2["p_lrt"][0] == "1.234747e-01"
    assert "%.6e" % np.nanmax(st2["p_lrt"]) == "9.997119e-01"
    assert "%.7g" % np.nanmax(st9["lambda_mle"]) == "0.7531109"
    assert "%.6e" % np.nanmax(st9["p_lrt"]) == "9.997119e-01"
    assert "%.0f" % float(bxd["kin_checksum"]) == "-116"  # test/dev_test_suite.sh:52
    assert st2.shape[0] == 7317 and bxd["U"].shape == (67, 67) and bxd["UtW"].shape == (67, 3)


def test_oracle_regenerates_fixture(oracle, bxd):
    """The committed SUMSTAT blocks are what the C restatement computes from the committed inputs."""
    X = bxd["X"].astype(np.float64)[:600]
    null = bxd["null"]
    for mode in (1, 2, 3, 4, 9):
        got = oracle.lmm_analyze(mode, bxd["U"], bxd["eval"], bxd["UtW"], bxd["Uty"], X,
                                 l_mle_null=null[0], logl_mle_H0=null[1])
        ref = bxd["stat_mode%d" % mode][:600]
        for k in ref.dtype.names:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-300, equal_nan=True, err_msg=k)
    l_mle, logl = oracle.calc_lambda_null("L", bxd["eval"], bxd["UtW"], bxd["Uty"])
    assert l_mle == pytest.approx(null[0], rel=1e-12) and logl == pytest.approx(null[1], rel=1e-12)


@pytest.mark.skipif(not os.path.exists(REF + "/example/BXD_geno.txt.gz"), reason="reference tree not mounted")
def test_bxd_end_to_end_from_reference_files(oracle):
    """Full restated pipeline from the reference's example files (build container only)."""
    ex = REF + "/example/"
    rs, G = oracle.read_bimbam_geno(ex + "BXD_geno.txt.gz")
    y, indp = oracle.read_pheno(ex + "BXD_pheno.txt")
    cvt, indc = oracle.read_cvt(ex + "BXD_covariates2.txt")
    ind, W = oracle.process_cvt_phen(indp, cvt, indc)
    assert ind.sum() == 67 and W.shape == (67, 3)
    isnp_k, _, _ = oracle.qc_snps(G, ind, W)
    assert isnp_k.sum() == 7317
    K = oracle.calc_kin(G[isnp_k == 1], 1)
    assert K.shape == (198, 198)  # "198" lines, test/dev_test_suite.sh:51
    isnp, _, _ = oracle.qc_snps(G, ind, W, maf_level=0.1)
    st, null, _ = oracle.run_lmm(2, G, ind, isnp, y, W, oracle.round10(K))
    assert "%.6e" % st["p_lrt"][0] == "1.234747e-01"
    assert "%.6e" % np.nanmax(st["p_lrt"]) == "9.997119e-01"


@pytest.mark.skipif(not os.path.exists(REF + "/test/data/issue188/2000.bed"), reason="reference tree not mounted")
def test_plink_decode_and_missing(oracle):
    """issue188 PLINK set with missing genotypes (test/dev_test_suite.sh:104-119): decode rules."""
    raw, ni_total, ph, ind = oracle.read_bed(REF + "/test/data/issue188/2000")
    assert ni_total == 1008 and raw.shape == (2000, 252)
    G = oracle.bed_decode(raw[:50], ni_total)
    vals = set(np.unique(G[~np.isnan(G)]).tolist())
    assert vals <= {0.0, 1.0, 2.0}
    # bit-level rule, src/lmm.cpp:1797-1812
    b = raw[3, 5]
    for j in range(4):
        v = (b >> (2 * j)) & 3
        exp = {0: 2.0, 2: 1.0, 3: 0.0}.get(v, np.nan)
        got = G[3, 20 + j]
        assert (np.isnan(got) and np.isnan(exp)) or got == exp


def test_lm_against_ols(oracle):
    """-lm restatement (src/lm.cpp:224-287,382-640) against an independent OLS fit: Wald p == two-sided t-test.
    (The reference's own -lm outputs on BXD and issue188 are checked in tests/test_reference_pin.py.)"""
    from scipy import stats
    rng = np.random.default_rng(1)
    n, c = 200, 3
    W = np.hstack([rng.standard_normal((n, 2)), np.ones((n, 1))])
    X = rng.integers(0, 3, size=(6, n)).astype(float)
    X[0, :7] = np.nan
    y = rng.standard_normal(n) + 0.3 * np.nan_to_num(X[1])
    out = oracle.lm_analyze(51, W, y, X)
    Xi = oracle.impute_mean(X)
    for s in range(6):
        A = np.hstack([W, Xi[s][:, None]])
        b = np.linalg.lstsq(A, y, rcond=None)[0]
        r = y - A @ b
        df = n - c - 1
        cov = (r @ r / df) * np.linalg.inv(A.T @ A)
        t = b[-1] / np.sqrt(cov[-1, -1])
        assert out["beta"][s] == pytest.approx(b[-1], rel=1e-11)
        assert out["se"][s] == pytest.approx(np.sqrt(cov[-1, -1]), rel=1e-11)
        assert out["p_wald"][s] == pytest.approx(2 * stats.t.sf(abs(t), df), rel=1e-10)


def test_gene_restatement_against_the_snp_path(oracle):
    """AnalyzeGene (src/lmm.cpp:1365-1471) only swaps roles: for every row y_g the alternative model is the one the SNP
    path fits with phenotype y_g and the fixed x as the single SNP, and (l_H0, logl_H0) is the null ML fit of (W, y_g).
    The restatement (with the reference's calc_null = false / zero-x-columns FUNC_PARAM) must reproduce both."""
    rng = np.random.default_rng(12)
    n, c, G = 120, 2, 9
    A = rng.standard_normal((n, n))
    K = A @ A.T / n
    U, ev, _ = oracle.eigen_decomp_zeroed(oracle.center_matrix(K))
    W = np.hstack([rng.standard_normal((n, c - 1)), np.ones((n, 1))])
    x = rng.integers(0, 3, size=n).astype(float)
    Y = rng.standard_normal((G, n)) + 0.5 * np.outer(rng.standard_normal(G), x)
    UtW, Utx = U.T @ W, U.T @ x
    out = oracle.gene_analyze(4, U, ev, UtW, Utx, Y)
    UtY = np.ascontiguousarray(Y @ U)  # the same rotated rows gene_analyze works on (lambda-hat is rounding sensitive)
    for g in range(G):
        Uty = UtY[g].copy()
        l0, logl0 = oracle.calc_lambda_null("L", ev, UtW, Uty)
        ref = oracle.lmm_batch_UtX(4, ev, UtW, Uty, Utx[None, :].copy(), l_mle_null=l0, logl_mle_H0=logl0)[0]
        for k in ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score", "logl_H1"):
            assert out[k][g] == pytest.approx(ref[k], rel=1e-9), (g, k)


def test_gxe_restatement_against_the_snp_path(oracle):
    """GXE (src/lmm.cpp:2283-2608) is the SNP path with per-SNP covariates [W, env, x_s] and tested variable x_s . env;
    logl_H0 is the null ML fit of those c + 2 covariates.  The restatement must reproduce the pinned SNP path run with
    that covariate matrix, SNP by SNP (beta with the 2 - x recoding sign)."""
    rng = np.random.default_rng(5)
    n, c, p = 110, 1, 7
    A = rng.standard_normal((n, n))
    U, ev, _ = oracle.eigen_decomp_zeroed(oracle.center_matrix(A @ A.T / n))
    W = np.ones((n, c))
    X = rng.integers(0, 3, size=(p, n)).astype(float)
    X[1] = 2 - (rng.random(n) < 0.15)          # a SNP with mean > 1 (gets recoded)
    X[2, :5] = np.nan
    env = rng.standard_normal(n)
    y = rng.standard_normal(n) + 0.3 * env
    UtW, Uty = U.T @ W, U.T @ y
    l0, _ = oracle.calc_lambda_null("L", ev, UtW, Uty)
    out = oracle.gxe_analyze(4, U, ev, UtW, Uty, env, X, l_mle_null=l0)
    Xi = oracle.impute_mean(X)
    for s in range(p):
        flip = np.nanmean(X[s]) > 1
        x = 2 - Xi[s] if flip else Xi[s]
        Xrot = np.ascontiguousarray(np.vstack([x, x * env]) @ U)  # rows: U^T x_s, U^T (x_s . env)
        UtWe = np.ascontiguousarray(np.hstack([UtW, (U.T @ env)[:, None], Xrot[0][:, None]]))
        lH0, loglH0 = oracle.calc_lambda_null("L", ev, UtWe, Uty)
        ref = oracle.lmm_batch_UtX(4, ev, UtWe, Uty, Xrot[1:2].copy(), l_mle_null=l0, logl_mle_H0=loglH0)[0]
        for k in ("se", "p_wald", "p_lrt", "p_score", "logl_H1"):
            assert out[k][s] == pytest.approx(ref[k], rel=1e-7), (s, k)
        assert out["beta"][s] == pytest.approx(-ref["beta"] if flip else ref["beta"], rel=1e-7), s
