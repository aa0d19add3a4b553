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
        mv.fit_null(c["ev"], np.ones((120, 1)), np.zeros((120, 9)))  # nine phenotypes (GEMMA_MV_DMAX = 8)
    assert e.value.code == L.EINVAL
    with pytest.raises(L.GemmaHipError) as e:
        mv.fit_null(c["ev"], np.ones((120, 12)), np.zeros((120, 2)))  # twelve covariates (+ the SNP > GEMMA_MV_CMAX = 12 rows)
    assert e.value.code == L.EINVAL


# ------------------------------------------------------------------ the run-time kernel (mvlmm_kernels_rt.hip)
@pytest.mark.parametrize("n,d,cw,p,seed,crt", [(300, 3, 1, 48, 5, 1), (257, 2, 2, 40, 6, 0), (350, 3, 6, 16, 22, 0)])
def test_run_time_kernel_equals_the_fixed_kernel(gpu_api, oracle, monkeypatch, n, d, cw, p, seed, crt):
    """GEMMA_HIP_MVLMM_RT=1 sends a shape that has a fixed kernel through the run-time one: same source (mvlmm.hip.h), same
    operation order per lane, different code generation (loops instead of unrolled register code: the compiler may contract
    other multiply-add pairs), so: equal to 1e-9 where both take the same number of EM iterations."""
    c = make_case(n, d, cw, p, seed)
    args = (c["U"], c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T), c["G"])
    a = gpu_api.MVLMM(a_mode=4, crt=crt, p_nr=0.5)
    fixed = a.AnalyzeBimbam(*args)
    monkeypatch.setenv("GEMMA_HIP_MVLMM_RT", "1")
    b = gpu_api.MVLMM(a_mode=4, crt=crt, p_nr=0.5)
    rt = b.AnalyzeBimbam(*args)
    for k in ("Vg_mle", "Ve_mle", "B_mle", "Vg_remle"):
        assert np.abs(a.null[k] - b.null[k]).max() < 1e-9 * np.abs(a.null[k]).max(), k
    _compare(rt, fixed, "rt vs fixed d=%d" % d, tight=1e-9)


@pytest.mark.parametrize("n,d,cw,p,seed,a_mode,crt", [(500, 6, 1, 24, 31, 4, 0), (420, 4, 5, 24, 32, 4, 1), (380, 2, 9, 32, 33, 4, 0),
                                                      (600, 7, 2, 12, 34, 1, 0), (450, 5, 4, 16, 35, 3, 1), (640, 8, 1, 8, 36, 1, 0)])
def test_analyze_beyond_the_fixed_kernels(gpu_api, oracle, n, d, cw, p, seed, a_mode, crt):
    """Shapes no fixed kernel is built for (d > 5; more than three covariates with d = 4, 5; more than six with d <= 3): the
    reference takes any (src/mvlmm.cpp:2972-3416); here the run-time kernel does, null block included (d > 4: pairwise
    two-trait initialisation)."""
    c = make_case(n, d, cw, p, seed)
    null, ref = _oracle_run(oracle, c, a_mode, X=c["G"], crt=crt, p_nr=0.5)
    mv = gpu_api.MVLMM(a_mode=a_mode, crt=crt, p_nr=0.5)
    got = mv.AnalyzeBimbam(c["U"], c["ev"], np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T), c["G"])
    for k in ("Vg_remle", "Ve_remle", "B_remle", "Vg_mle", "Ve_mle", "B_mle"):
        assert np.abs(mv.null[k] - null[k]).max() < 1e-6 * np.abs(null[k]).max(), k
    assert mv.null["logl_mle"] == pytest.approx(null["logl_mle"], rel=1e-10)
    _compare(got, ref, "rt d=%d c=%d" % (d, cw))


@pytest.mark.parametrize("n,d,cw,p,seed,a_mode,crt,plink", [(300, 2, 1, 40, 41, 4, 0, False), (320, 3, 2, 24, 42, 4, 1, False),
                                                            (280, 2, 1, 32, 43, 1, 0, True), (300, 1, 1, 32, 44, 4, 0, False)])
def test_analyze_gxe(gpu_api, oracle, n, d, cw, p, seed, a_mode, crt, plink):
    """MVLMM::AnalyzeBimbamGXE / AnalyzePlinkGXE (src/mvlmm.cpp:3970-4870) against orc_mvlmm_batch_gxe (pinned on the reference's
    own -gxe output in tests/test_reference_pin.py): null fit on (W, env), per-SNP null on (W, env, x), tested row x o env, alleles
    switched where the mean exceeds 1 (beta changes sign)."""
    c = make_case(n, d, cw, p, seed)
    rng = np.random.default_rng(seed + 100)
    env = rng.standard_normal(n)
    U = c["U"]
    G = c["G"].copy()
    G[::3] = 2.0 - G[::3]  # a third of the SNPs with the other allele counted: mean > 1 -> the switch of :4232-4236
    G[rng.random(G.shape) < 0.02] = np.nan
    if plink:
        G = np.where(np.isnan(G), np.nan, np.round(G))
    X = oracle.impute_mean(G)
    flip = X.mean(1) > 1
    X[flip] = 2.0 - X[flip]
    W_env = np.ascontiguousarray(np.vstack([c["UtW"], (U.T @ env)[None, :]]))
    cfg = oracle.mv_cfg(crt=crt, p_nr=0.5)
    null = oracle.mvlmm_null(cfg, c["ev"], W_env, c["UtY"])
    ref = oracle.mvlmm_batch_gxe(a_mode, cfg, c["ev"], W_env, c["UtY"], np.ascontiguousarray(X @ U),
                                 np.ascontiguousarray((X * env[None, :]) @ U), null)
    ref["beta"][flip] *= -1.0
    assert flip.sum() >= 3
    mv = gpu_api.MVLMM(a_mode=a_mode, crt=crt, p_nr=0.5)
    UtW, UtY = np.ascontiguousarray(c["UtW"].T), np.ascontiguousarray(c["UtY"].T)
    if plink:
        # .bed codes per individual (src/lmm.cpp:1797-1812): 0 -> 2, 1 -> NA, 2 -> 1, 3 -> 0
        codes = np.where(np.isnan(G), 1, np.where(G == 2, 0, np.where(G == 1, 2, 3))).astype(np.uint8)
        pad = np.zeros((p, (n + 3) // 4 * 4), dtype=np.uint8)
        pad[:, :n] = codes
        raw = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
        assert np.array_equal(np.nan_to_num(oracle.bed_decode(raw, n), nan=-1), np.nan_to_num(G, nan=-1))
        got = mv.AnalyzePlinkGXE(U, c["ev"], UtW, UtY, env, raw, np.ones(n, dtype=np.int32))
    else:
        got = mv.AnalyzeBimbamGXE(U, c["ev"], UtW, UtY, env, G)
    _compare(got, ref, "gxe d=%d" % d)
