"""The multivariate-LMM oracle (oracle/mvlmm_oracle.c) has no reference golden to pin it (the reference's own tests
check line counts only, test/dev_test_suite.sh:196-208, and the eigenvector file of that run is not in the tree), so it
is checked from three independent sides: finite differences of its own log-likelihood, the univariate oracle (which IS
pinned on the reference's goldens) at d = 1, and a second formulation of the same algorithm -- the kernel source
gemma_amd/csrc/mvlmm.hip.h compiled for one CPU lane (tests/host/mvlmm_harness.cpp), which works in the simultaneously
diagonalising basis with moment tables where the oracle keeps explicit H_k^-1 blocks and the full Q."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def make_case(n, d, cw, p, seed):
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.1, 0.5, 400)
    Gk = rng.binomial(2, maf[None, :], size=(n, 400)).astype(float)
    Gk -= Gk.mean(0)
    K = Gk @ Gk.T / 400
    ev, U = np.linalg.eigh(K)
    ev[ev < 1e-10] = 0
    Lg = np.tril(rng.standard_normal((d, d))) * 0.25 + np.eye(d) * 1.3
    Le = np.tril(rng.standard_normal((d, d))) * 0.25 + np.eye(d)
    Gs = rng.binomial(2, 0.3, size=(p, n)).astype(float)
    Y = (np.linalg.cholesky(K + 1e-8 * np.eye(n)) @ rng.standard_normal((n, d))) @ Lg.T + rng.standard_normal((n, d)) @ Le.T
    hit = min(3, p - 1)
    Y[:, 0] += 0.7 * Gs[hit]
    if d > 1:
        Y[:, 1] -= 0.6 * Gs[hit]
    W = np.column_stack([np.ones(n)] + [rng.standard_normal(n) for _ in range(cw - 1)])
    return {"U": U, "ev": ev, "W": W, "Y": Y, "G": Gs, "UtW": np.ascontiguousarray((U.T @ W).T),
            "UtY": np.ascontiguousarray((U.T @ Y).T), "UtX": np.ascontiguousarray(Gs @ U)}


@pytest.mark.parametrize("nu", [1, 2, 3, 4, 5, 6])
def test_chisq_upper_tail(nu):
    from scipy.stats import chi2
    for x in (1e-8, 0.3, 1.0, 4.7, 19.0, 80.0, 400.0):
        assert O.chisq_Q(x, nu) == pytest.approx(chi2.sf(x, nu), rel=2e-13, abs=1e-300)
    assert O.chisq_Q(-1.0, nu) == 1.0 and O.chisq_Q(0.0, nu) == 1.0


@pytest.mark.parametrize("func", ["R", "L"])
def test_gradient_and_hessian_are_derivatives_of_logl(func):
    """CalcDev's gradient / Hessian (src/mvlmm.cpp:2360-2554) against central differences of the logl MphNR maximises.
    The Hessian check covers both halves of the V_g / V_e cross block, which :2494-2504 fills from one ordering only."""
    c = make_case(220, 3, 2, 2, 11)
    d = 3
    X = np.vstack([c["UtW"], c["UtX"][:1]])
    rng = np.random.default_rng(1)
    A, B = rng.standard_normal((d, d)), rng.standard_normal((d, d))
    Vg, Ve = A @ A.T / d + 0.3 * np.eye(d), B @ B.T / d + 0.5 * np.eye(d)
    ll, g, H = O.mph_dev(func, c["ev"], X, c["UtY"], Vg, Ve)
    idx = [(i, j) for i in range(d) for j in range(i, d)]
    h = 1e-6

    def pert(which, i, j, s):
        P, Q = Vg.copy(), Ve.copy()
        M = P if which == 0 else Q
        M[i, j] += s
        if i != j:
            M[j, i] += s
        return P, Q

    gfd, Hfd = np.zeros_like(g), np.zeros_like(H)
    for q, (which, (i, j)) in enumerate([(w, ij) for w in (0, 1) for ij in idx]):
        lp = O.mph_dev(func, c["ev"], X, c["UtY"], *pert(which, i, j, h))
        lm = O.mph_dev(func, c["ev"], X, c["UtY"], *pert(which, i, j, -h))
        gfd[q] = (lp[0] - lm[0]) / (2 * h)
        Hfd[:, q] = (lp[1] - lm[1]) / (2 * h)
    assert np.abs(g - gfd).max() < 2e-6 * np.abs(g).max()
    assert np.abs(H - Hfd).max() < 1e-7 * np.abs(H).max()


def test_em_increases_the_likelihood_and_nr_finds_a_stationary_point():
    d, interior = 3, 0
    for seed in (12, 21, 22, 23):
        c = make_case(250, d, 1, 1, seed)
        for func in "RL":
            Vg, Ve, B = np.eye(d) * 0.5, np.eye(d) * 0.7, np.zeros((d, 1))
            lls = []
            for it in (2, 4, 8, 16, 64):
                vg, ve, b = Vg.copy(), Ve.copy(), B.copy()
                lls.append(O.mph_em(func, it, 0.0, c["ev"], c["UtW"], c["UtY"], vg, ve, b))
            assert all(b2 >= a2 - 1e-9 for a2, b2 in zip(lls, lls[1:])), lls
            ll_nr, Hi = O.mph_nr(func, 100, 1e-10, c["ev"], c["UtW"], c["UtY"], vg, ve)
            assert ll_nr >= lls[-1] - 1e-9
            if min(np.linalg.eigvalsh(vg).min(), np.linalg.eigvalsh(ve).min()) < 0.02:
                continue  # maximum on the boundary of the positive-definite cone: MphNR stops at its step-halving limit
            interior += 1
            _, g, H = O.mph_dev(func, c["ev"], c["UtW"], c["UtY"], vg, ve)
            assert np.abs(g).max() < 1e-4  # stationary
            assert np.all(np.linalg.eigvalsh((H + H.T) / 2) < 0)  # a maximum
            assert np.all(np.diag(Hi) > 0)  # -H^-1 is the variance matrix (:2742-2744)
    assert interior >= 2


def test_one_trait_reduces_to_the_univariate_lmm():
    """d = 1: V_g / V_e is the REML lambda of -lmm, beta and Vbeta of MphCalcP are CalcRLWald's beta and se^2."""
    c = make_case(300, 1, 2, 25, 13)
    cfg = O.mv_cfg(em_prec=1e-10, nr_prec=1e-12)
    null = O.mvlmm_null(cfg, c["ev"], c["UtW"], c["UtY"])
    UtWn = np.ascontiguousarray(c["UtW"].T)
    lam, _ = O.calc_lambda_null("R", c["ev"], UtWn, c["UtY"][0])
    assert null["Vg_remle"][0, 0] / null["Ve_remle"][0, 0] == pytest.approx(lam, rel=2e-5)
    lam_m, _ = O.calc_lambda_null("L", c["ev"], UtWn, c["UtY"][0])
    assert null["Vg_mle"][0, 0] / null["Ve_mle"][0, 0] == pytest.approx(lam_m, rel=2e-5)
    uni = O.lmm_batch_UtX(1, c["ev"], UtWn, c["UtY"][0], c["UtX"])
    X = np.vstack([c["UtW"], c["UtX"][:1]])
    for s in range(c["UtX"].shape[0]):
        X[-1] = c["UtX"][s]
        vg, ve, b = null["Vg_mle"].copy(), null["Ve_mle"].copy(), np.zeros((1, 3))
        O.mph_em("R", 10000, 1e-10, c["ev"], X, c["UtY"], vg, ve, b)
        O.mph_nr("R", 100, 1e-12, c["ev"], X, c["UtY"], vg, ve)
        p, beta, Vbeta = O.mph_calcp(c["ev"], c["UtX"][s], c["UtW"], c["UtY"], vg, ve)
        assert vg[0, 0] / ve[0, 0] == pytest.approx(uni["lambda_remle"][s], rel=1e-3, abs=2e-5)
        # the univariate lambda carries ~5 digits (Newton stops at 1e-5 relative, src/lmm.cpp:2073): compare on the s.e. scale
        assert abs(beta[0] - uni["beta"][s]) < 1e-4 * uni["se"][s]
        assert np.sqrt(Vbeta[0, 0]) == pytest.approx(uni["se"][s], rel=1e-4)


# ------------------------------------------------------------------ the kernel source on one CPU lane
class MvArgs(C.Structure):
    dp = C.POINTER(C.c_double)
    _fields_ = [("UtX", dp), ("ld", C.c_long), ("l", C.c_long), ("n", C.c_int), ("eval", dp), ("Wt", dp), ("Yt", dp),
                ("Vg_null", C.c_double * 64), ("Ve_null", C.c_double * 64), ("B_null", C.c_double * 96),  # MV_DMAX = 8, MV_CMAX = 12
                ("logl_H0", C.c_double), ("a_mode", C.c_int), ("em_iter", C.c_int), ("em_prec", C.c_double),
                ("nr_iter", C.c_int), ("nr_prec", C.c_double), ("p_nr", C.c_double), ("out", dp), ("stride", C.c_int),
                ("crt", C.c_int), ("d", C.c_int), ("c", C.c_int), ("UtX2", dp), ("flip", C.POINTER(C.c_int)), ("scratch", dp)]


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(HERE, "host", "mvlmm_harness.cpp")
    so = os.path.join(HERE, "host", "libmvlmm_harness.so")
    hdr = os.path.join(HERE, "..", "gemma_amd", "csrc", "mvlmm.hip.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    H = C.CDLL(so)
    H.mvh_args_size.restype = C.c_size_t
    assert H.mvh_args_size() == C.sizeof(MvArgs)
    return H


@pytest.mark.parametrize("n,d,cw,p,seed,crt", [(300, 3, 1, 40, 5, 0), (257, 2, 2, 30, 6, 0), (200, 1, 1, 30, 7, 0), (400, 4, 1, 12, 8, 0),
                                               (600, 5, 2, 6, 9, 0), (350, 3, 3, 10, 10, 0),
                                               (300, 3, 1, 40, 5, 1), (257, 2, 2, 30, 6, 1), (200, 1, 1, 30, 7, 1),
                                               (400, 4, 1, 12, 8, 1), (350, 3, 3, 10, 10, 1),
                                               (300, 3, 1, 40, 5, 2), (257, 2, 2, 30, 6, 2),
                                               (320, 2, 4, 16, 21, 0), (350, 3, 6, 10, 22, 0), (300, 1, 5, 12, 23, 2), (340, 3, 5, 10, 28, 1)])
def test_kernel_source_on_one_lane_matches_the_oracle(harness, n, d, cw, p, seed, crt):
    """crt = 1: the reference's -crt.  The kernel source computes CalcCRT's traces in the rotated basis from its moment tables
    (MvNr::crt_factors), the oracle as the reference does, from dense dc x dc products (pinned on the reference's own crt_a, b, c
    in tests/test_reference_pin.py): two formulations, same corrected p values."""
    c = make_case(n, d, cw, p, seed)
    p_nr = 1e-3
    if crt == 2: # every second SNP goes through MphNR and PCRT in all three modes (score included)
        crt, p_nr = 1, 0.5
    cfg = O.mv_cfg(crt=crt, p_nr=p_nr)
    null = O.mvlmm_null(cfg, c["ev"], c["UtW"], c["UtY"])
    # an interior null fit: on the boundary of the positive-definite cone (an eigenvalue of V_e or V_g ~ 1e-8) every
    # downstream number is conditioned like 1e8 and two correct formulations agree to a few digits only
    assert min(np.linalg.eigvalsh(null["Ve_mle"]).min(), np.linalg.eigvalsh(null["Vg_mle"]).min()) > 1e-3
    ref = O.mvlmm_batch(4, cfg, c["ev"], c["UtW"], c["UtY"], c["UtX"], null)
    v = d * (d + 1) // 2
    stride = 3 * v + d + 3
    out = np.zeros((p, stride))
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    a = MvArgs()
    a.UtX, a.ld, a.l, a.n = P(c["UtX"]), n, p, n
    a.eval, a.Wt, a.Yt = P(c["ev"]), P(c["UtW"]), P(c["UtY"])
    for i, x in enumerate(null["Vg_mle"].ravel()):
        a.Vg_null[i] = x
    for i, x in enumerate(null["Ve_mle"].ravel()):
        a.Ve_null[i] = x
    for i, x in enumerate(null["B_mle"].ravel()):
        a.B_null[i] = x
    a.logl_H0, a.a_mode = null["logl_mle"], 4
    a.em_iter, a.em_prec, a.nr_iter, a.nr_prec, a.p_nr = 1000, 1e-3, 10, 1e-3, p_nr
    a.out, a.stride = P(out), stride
    a.crt = crt
    assert harness.mvh_batch(d, cw + 1, C.byref(a)) == 0
    got = {"beta": out[:, :d], "Vbeta": out[:, d:d + v], "Vg": out[:, d + v:d + 2 * v], "Ve": out[:, d + 2 * v:d + 3 * v],
           "p_wald": out[:, d + 3 * v], "p_lrt": out[:, d + 3 * v + 1], "p_score": out[:, d + 3 * v + 2]}
    assert (ref["p_wald"] < 1e-3).sum() >= 1  # the Newton-Raphson branch is exercised
    if p_nr > 0.1:
        assert (ref["p_score"] < p_nr).sum() >= 5
    for k in got:
        rel = np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300)
        assert rel.max() < 1e-8, (k, rel.max())


def _harness_batch(harness, c, null, d, cw, p_nr, crt, fixed, a_mode=4, utx2=None, w=None):
    n, p = c["UtX"].shape[1], c["UtX"].shape[0]
    v = d * (d + 1) // 2
    stride = 3 * v + d + 3
    out = np.zeros((p, stride))
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    W = c["UtW"] if w is None else w
    a = MvArgs()
    a.UtX, a.ld, a.l, a.n = P(c["UtX"]), n, p, n
    a.eval, a.Wt, a.Yt = P(c["ev"]), P(W), P(c["UtY"])
    for i, x in enumerate(null["Vg_mle"].ravel()):
        a.Vg_null[i] = x
    for i, x in enumerate(null["Ve_mle"].ravel()):
        a.Ve_null[i] = x
    for i, x in enumerate(null["B_mle"].ravel()):
        a.B_null[i] = x
    a.logl_H0, a.a_mode = null["logl_mle"], a_mode
    a.em_iter, a.em_prec, a.nr_iter, a.nr_prec, a.p_nr = 1000, 1e-3, 10, 1e-3, p_nr
    a.out, a.stride = P(out), stride
    a.crt = crt
    rows = W.shape[0] + 1
    if utx2 is not None:
        a.UtX2 = P(utx2)
        rows += 1
    assert harness.mvh_batch2(d, rows, C.byref(a), fixed) == 0
    return {"beta": out[:, :d], "Vbeta": out[:, d:d + v], "Vg": out[:, d + v:d + 2 * v], "Ve": out[:, d + 2 * v:d + 3 * v],
            "p_wald": out[:, d + 3 * v], "p_lrt": out[:, d + 3 * v + 1], "p_score": out[:, d + 3 * v + 2]}


@pytest.mark.parametrize("n,d,cw,p,seed,crt", [(300, 3, 1, 24, 5, 1), (257, 2, 2, 20, 6, 0), (400, 4, 1, 8, 8, 1), (350, 3, 6, 8, 22, 0)])
def test_run_time_instance_is_the_fixed_instance_bit_for_bit(harness, n, d, cw, p, seed, crt):
    """One source, two forms (mvlmm.hip.h: DT, CT > 0 fixed, 0 run-time): on one CPU lane and without fused multiply-adds the same
    operations run in the same order, so where a fixed instance exists the run-time one must return the same bits."""
    c = make_case(n, d, cw, p, seed)
    null = O.mvlmm_null(O.mv_cfg(crt=crt, p_nr=0.5), c["ev"], c["UtW"], c["UtY"])
    a = _harness_batch(harness, c, null, d, cw, 0.5, crt, 1)
    b = _harness_batch(harness, c, null, d, cw, 0.5, crt, 0)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("n,d,cw,p,seed,crt", [(500, 6, 1, 6, 31, 0), (420, 4, 5, 8, 32, 1), (380, 2, 9, 10, 33, 0), (600, 7, 2, 4, 34, 0),
                                               (450, 5, 4, 6, 35, 1)])
def test_run_time_instance_matches_the_oracle_beyond_the_fixed_kernels(harness, n, d, cw, p, seed, crt):
    """(d, c) for which no fixed kernel is built (d > 5, or more covariates than 3 for d = 4, 5 / 6 for d <= 3): the reference takes
    any (src/mvlmm.cpp:2972-3416)."""
    c = make_case(n, d, cw, p, seed)
    cfg = O.mv_cfg(crt=crt, p_nr=0.5)
    null = O.mvlmm_null(cfg, c["ev"], c["UtW"], c["UtY"])
    assert min(np.linalg.eigvalsh(null["Ve_mle"]).min(), np.linalg.eigvalsh(null["Vg_mle"]).min()) > 1e-3
    ref = O.mvlmm_batch(4, cfg, c["ev"], c["UtW"], c["UtY"], c["UtX"], null)
    got = _harness_batch(harness, c, null, d, cw, 0.5, crt, 0)
    for k in got:
        rel = np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300)
        assert rel.max() < 1e-8, (k, rel.max())


@pytest.mark.parametrize("n,d,cw,p,seed,a_mode,crt", [(300, 2, 1, 10, 41, 4, 0), (320, 3, 2, 8, 42, 4, 1), (280, 2, 1, 12, 43, 1, 0),
                                                      (300, 1, 1, 10, 44, 4, 0)])
def test_gxe_on_one_lane_matches_the_oracle(harness, n, d, cw, p, seed, a_mode, crt):
    """The interaction test (MVLMM::AnalyzeBimbamGXE, src/mvlmm.cpp:3970-4414): per-SNP null fits on (W, env, x), tested row x o env;
    kernel source (mv_one_snp_gxe, run-time instance) against the oracle's restatement of the reference's loop."""
    c = make_case(n, d, cw, p, seed)
    rng = np.random.default_rng(seed + 100)
    env = rng.standard_normal(n)
    U = c["U"]
    W_env = np.ascontiguousarray(np.vstack([c["UtW"], (U.T @ env)[None, :]]))
    Gs = c["G"]
    UtX2 = np.ascontiguousarray((Gs * env[None, :]) @ U)
    cfg = O.mv_cfg(crt=crt, p_nr=0.5)
    null = O.mvlmm_null(cfg, c["ev"], W_env, c["UtY"])
    ref = O.mvlmm_batch_gxe(a_mode, cfg, c["ev"], W_env, c["UtY"], c["UtX"], UtX2, null)
    got = _harness_batch(harness, c, null, d, cw + 1, 0.5, crt, 0, a_mode=a_mode, utx2=UtX2, w=W_env)
    for k in got:
        rel = np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300)
        assert rel.max() < 1e-8, (k, rel.max())


class MvNullArgs(C.Structure):
    _fields_ = [("g", MvArgs), ("em_iter", C.c_int), ("em_prec", C.c_double), ("Vg0", C.c_double * 64), ("Ve0", C.c_double * 64),
                ("out", C.POINTER(C.c_double))]


@pytest.mark.parametrize("n,d,cw,seed", [(260, 6, 2, 51), (240, 3, 8, 52), (300, 2, 1, 53)])
def test_null_fit_of_the_run_time_instance_matches_the_oracle(harness, n, d, cw, seed):
    """mv_null_fit<0, 0> (EM + Newton-Raphson for REML, then ML from the REML fit, GLS B) on one CPU lane from the oracle's own
    starting point (MphInitial) against orc_mph_em / orc_mph_nr: the null block of shapes that have no fixed kernel."""
    c = make_case(n, d, cw, 2, seed)
    cfg = O.mv_cfg()
    ref = O.mvlmm_null(cfg, c["ev"], c["UtW"], c["UtY"])
    Vg0, Ve0, _ = O.mph_initial(cfg, c["ev"], c["UtW"], c["UtY"])
    harness.mvh_null_args_size.restype = C.c_size_t
    assert harness.mvh_null_args_size() == C.sizeof(MvNullArgs)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    a = MvNullArgs()
    a.g.n, a.g.eval, a.g.Wt, a.g.Yt = n, P(c["ev"]), P(c["UtW"]), P(c["UtY"])
    a.g.nr_iter, a.g.nr_prec, a.g.d, a.g.c = cfg.nr_iter, cfg.nr_prec, d, cw
    a.em_iter, a.em_prec = cfg.em_iter, cfg.em_prec
    for i, x in enumerate(np.asarray(Vg0).ravel()):
        a.Vg0[i] = x
    for i, x in enumerate(np.asarray(Ve0).ravel()):
        a.Ve0[i] = x
    blk = 2 * d * d + d * cw + 1
    out = np.zeros(2 * blk)
    a.out = P(out)
    assert harness.mvh_null_rt(C.byref(a)) == 0
    for p, tag in ((0, "remle"), (1, "mle")):
        o = out[p * blk:(p + 1) * blk]
        assert np.abs(o[:d * d].reshape(d, d) - ref["Vg_" + tag]).max() < 1e-7 * np.abs(ref["Vg_" + tag]).max()
        assert np.abs(o[d * d:2 * d * d].reshape(d, d) - ref["Ve_" + tag]).max() < 1e-7 * np.abs(ref["Ve_" + tag]).max()
        assert np.abs(o[2 * d * d:2 * d * d + d * cw].reshape(d, cw) - ref["B_" + tag]).max() < 1e-7 * np.abs(ref["B_" + tag]).max()
        assert o[-1] == pytest.approx(ref["logl_" + tag], rel=1e-10)
