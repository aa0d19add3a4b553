"""Multivariate LMM on the GPU (gemma_hip_mvlmm_*: one SNP per wavefront, csrc/mvlmm.hip.h) against oracle/mvlmm_oracle.c.
The per-SNP EM stops when |logl_new - logl_old| < 1e-3 (src/mvlmm.cpp:667 with em_prec * 10), so an iteration count can
differ between two correct implementations when a step lands within rounding of that threshold; the estimates then move
by ~1e-4.  Criterion (as for the univariate lambda): >= 97 % of the SNPs within 1e-6 relative on every field,
all of them within 5e-3."""
import numpy as np
import pytest

from test_oracle_mvlmm import make_case

pytestmark = pytest.mark.gpu

FIELDS = ("beta", "Vbeta", "Vg", "Ve", "p_wald", "p_lrt", "p_score")


def _compare(got, ref, tag, tight=1e-6, loose=5e-3, frac=0.97):
    l = ref["p_wald"].shape[0]
    bad = np.zeros(l, dtype=bool)
    for k in FIELDS:
        g, r = np.asarray(got[k]).reshape(l, -1), np.asarray(ref[k]).reshape(l, -1)
        scale = np.maximum(np.abs(r).max(axis=1, keepdims=True), 1e-300)
        rel = (np.abs(g - r) / scale).max(axis=1)
        assert np.all(np.isfinite(g)), (tag, k)
        assert rel.max() < loose, (tag, k, rel.max())
        bad |= rel > tight
    assert bad.mean() <= 1 - frac, (tag, bad.mean())


def _oracle_run(oracle, c, a_mode, X=None, **cfg_kw):
    cfg = oracle.mv_cfg(**cfg_kw)
    null = oracle.mvlmm_null(cfg, c["ev"], c["UtW"], c["UtY"])
    UtX = c["UtX"] if X is None else np.ascontiguousarray(X @ c["U"])
    return null, oracle.mvlmm_batch(a_mode, cfg, c["ev"], c["UtW"], c["UtY"], UtX, null)


@pytest.mark.parametrize("n,d,cw,seed", [(320, 2, 4, 21), (350, 3, 6, 22), (300, 1, 5, 23),  # four to six covariates: d <= 3 (round 3)
                                        (300, 3, 1, 5), (257, 2, 2, 6), (200, 1, 1, 7), (400, 4, 1, 8), (600, 5, 2, 9),
                                         (350, 3, 3, 10)])
def test_null_model_block(gpu_api, oracle, n, d, cw, seed):
    """MphInitial + MphEM + MphNR + MphCalcBeta for 'R' then 'L' (src/mvlmm.cpp:3056-3208); d = 5 takes the two-trait
    initialisation of :2805-2884."""
    c = make_case(n, d, cw, 4, seed)
    ref = oracle.mvlmm_null(oracle.mv_cfg(), c["ev"], c["UtW"], c["UtY"])
    mv = gpu_api.MVLMM(a_mode=1)
    got = mv.fit_null(c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T))
    for k in ("Vg_remle", "Ve_remle", "B_remle", "Vg_mle", "Ve_mle", "B_mle"):
        assert np.abs(got[k] - ref[k]).max() < 1e-6 * np.abs(ref[k]).max(), k
    assert got["logl_remle"] == pytest.approx(ref["logl_remle"], rel=1e-10)
    assert got["logl_mle"] == pytest.approx(ref["logl_mle"], rel=1e-10)


@pytest.mark.parametrize("n,d,cw,p,seed,a_mode", [(300, 3, 1, 300, 5, 4), (257, 2, 2, 130, 6, 4), (200, 1, 1, 70, 7, 4),
                                                  (400, 4, 1, 40, 8, 1), (600, 5, 2, 24, 9, 2), (350, 3, 3, 50, 10, 3),
                                                  (300, 3, 1, 64, 15, 1), (300, 3, 2, 64, 16, 2),
                                                  (320, 2, 4, 48, 21, 4), (350, 3, 6, 24, 22, 4), (340, 3, 5, 24, 28, 1)])
def test_analyze_bimbam(gpu_api, oracle, n, d, cw, p, seed, a_mode):
    c = make_case(n, d, cw, p, seed)
    G = c["G"].copy()
    rng = np.random.default_rng(seed)
    G[rng.random(G.shape) < 0.02] = np.nan  # NA genotypes: imputed with the SNP mean (src/mvlmm.cpp:3254-3261)
    Gi = oracle.impute_mean(G)
    null, ref = _oracle_run(oracle, c, a_mode, X=Gi)
    mv = gpu_api.MVLMM(a_mode=a_mode)
    got = mv.AnalyzeBimbam(c["U"], c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T), G)
    if a_mode in (1, 4):
        assert (ref["p_wald"] < 1e-3).sum() >= 1  # Newton-Raphson refinement taken
    _compare(got, ref, "bimbam d=%d" % d)


@pytest.mark.parametrize("n,d,cw,p,seed,p_nr", [(300, 3, 1, 120, 5, 1e-3), (257, 2, 2, 96, 6, 0.5), (400, 4, 1, 40, 8, 0.5),
                                                (350, 3, 3, 48, 10, 0.5), (600, 5, 2, 16, 9, 0.5)])
def test_analyze_bimbam_with_crt(gpu_api, oracle, n, d, cw, p, seed, p_nr):
    """-crt (src/mvlmm.cpp:2054-2331 CalcCRT, :2952-2970 PCRT): the SNPs that reach MphNR get Edgeworth-corrected p values.  The
    kernel takes the correction factors from its moment tables in the rotated basis (MvNr::crt_factors), the oracle from dense
    products as the reference does (its crt_a, b, c are pinned on the reference's in tests/test_reference_pin.py).  p_nr = 0.5
    sends about half of the SNPs down that road, in all three modes (score: one CalcDev at the null estimates)."""
    c = make_case(n, d, cw, p, seed)
    null, ref = _oracle_run(oracle, c, 4, X=c["G"], crt=1, p_nr=p_nr)
    _, plain = _oracle_run(oracle, c, 4, X=c["G"], crt=0, p_nr=p_nr)
    changed = sum(int((np.abs(ref[k] - plain[k]) > 1e-9 * plain[k]).sum()) for k in ("p_wald", "p_lrt", "p_score"))
    assert changed >= (1 if p_nr < 0.1 else p // 2)  # the correction is not a no-op on this case
    mv = gpu_api.MVLMM(a_mode=4, crt=1, p_nr=p_nr)
    got = mv.AnalyzeBimbam(c["U"], c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T), c["G"])
    _compare(got, ref, "crt d=%d" % d)


def test_analyze_plink(gpu_api, oracle):
    """AnalyzePlink (src/mvlmm.cpp:3418-3899): 2-bit rows over ni_total individuals, indicator_idv drops some."""
    from test_gpu_parity import _plink_case
    rng = np.random.default_rng(77)
    ni_total, p, d = 420, 96, 3
    ind, raw = _plink_case(oracle, rng, ni_total, p)
    n = int(ind.sum())
    X = oracle.impute_mean(oracle.bed_decode(raw, ni_total, ind))
    c = make_case(n, d, 1, 4, 78)
    null, ref = _oracle_run(oracle, c, 4, X=X)
    mv = gpu_api.MVLMM(a_mode=4)
    got = mv.AnalyzePlink(c["U"], c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T), raw, ind)
    _compare(got, ref, "plink")


def test_state_and_argument_errors(gpu_api, oracle):
    from gemma_amd import _lib as L
    lib = L.lib()
    out = np.zeros(64)
    x = np.zeros((1, 10))
    assert lib.gemma_hip_mvlmm_batch(L.GENO_F64_SNP_MAJOR, x.ctypes.data, 1, 10, out.ctypes.data) == L.ESTATE
    c = make_case(120, 2, 1, 2, 3)
    mv = gpu_api.MVLMM()
    with pytest.raises(L.GemmaHipError) as e:
        mv.fit_null(c["ev"], np.ones((120, 1)), np.zeros((120, 6)))  # six phenotypes
    assert e.value.code == L.EINVAL
    with pytest.raises(L.GemmaHipError) as e:
        mv.fit_null(c["ev"], np.ones((120, 7)), np.zeros((120, 2)))  # seven covariates (d <= 3: up to six)
    assert e.value.code == L.EINVAL
    with pytest.raises(L.GemmaHipError) as e:
        mv.fit_null(c["ev"], np.ones((120, 4)), np.zeros((120, 4)))  # four covariates with four phenotypes (d > 3: up to three)
    assert e.value.code == L.EINVAL
