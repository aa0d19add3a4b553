"""The two-tier lambda-hat criterion of the GPU parity tests (tests/test_gpu_parity.py: _cmp_stats / _classify_lambda;
SURVEY App. A.5), exercised on the CPU: it must accept what a flipped Brent / Newton trip count does to
lambda-hat and reject a wrong value.  "Flipped" is produced here the honest way: the same C restatement compiled with
another floating-point arithmetic (objects built with -O3 -ffast-math -march=native: reassociated sums, FMA contraction), which moves dev1
at the 1e-13 level exactly as another summation order on the GPU does."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def parity():
    import test_gpu_parity as T
    return T


@pytest.fixture(scope="module")
def other_arithmetic(tmp_path_factory):
    """orc_lmm_batch of the oracle's sources under another arithmetic, as a callable (a_mode, ev, UtW, Uty, UtX, null) -> SUMSTAT"""
    import ctypes as C
    from oracle import oracle as O
    so = str(tmp_path_factory.mktemp("fast") / "liboracle_fast.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("gemma_oracle.c", "mvlmm_oracle.c")]
    # compiled and linked in two steps: -ffast-math at LINK time would pull in crtfastmath.o, whose constructor switches the
    # loading process to flush-to-zero -- the other CPU tests of this session must keep IEEE denormals
    objs = []
    for f in src:
        o = so + "." + os.path.basename(f) + ".o"
        subprocess.check_call(["gcc", "-O3", "-ffast-math", "-march=native", "-fPIC", "-std=gnu99", "-w", "-c", f, "-o", o])
        objs.append(o)
    subprocess.check_call(["gcc", "-shared", "-o", so] + objs + ["-lm"])
    L = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    L.orc_lmm_batch.argtypes = [C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_size_t, C.c_double, C.c_double, C.c_size_t,
                                C.c_double, C.c_double, C.c_int, dp, C.POINTER(O.SumStat), C.POINTER(C.c_long)]

    def run(a_mode, ev, UtW, Uty, UtX, l_mle_null, logl_mle_H0):
        ev, UtW, Uty, UtX = (np.ascontiguousarray(a, dtype=np.float64) for a in (ev, UtW, Uty, UtX))
        out = np.zeros(UtX.shape[0], dtype=O.SUMSTAT_DTYPE)
        cr = np.zeros(2)
        L.orc_lmm_batch(a_mode, UtW.shape[0], UtW.shape[1], O._dp(ev), O._dp(UtW), O._dp(Uty), O._dp(UtX), UtX.shape[0], 1e-5, 1e5, 10,
                        l_mle_null, logl_mle_H0, 0, O._dp(cr), out.ctypes.data_as(C.POINTER(O.SumStat)), None)
        return out
    return run


def test_flipped_trip_counts_are_accepted_on_bxd(parity, oracle, bxd, other_arithmetic, monkeypatch):
    """BXD (n = 67: flat optima, the hardest case): another arithmetic moves ~1 % of the lambda-hats by up to 1.3e-4; every one
    of them must classify as a flipped trip count, none as wrong."""
    monkeypatch.setattr(parity, "_record", lambda line: None)
    U, ev, UtW, Uty, X = bxd["U"], bxd["eval"], bxd["UtW"], bxd["Uty"], bxd["X"].astype(np.float64)
    null = bxd["null"]
    UtX = np.ascontiguousarray(oracle.impute_mean(X) @ U)
    ref = oracle.lmm_batch_UtX(4, ev, UtW, Uty, UtX, l_mle_null=null[0], logl_mle_H0=null[1])
    var = other_arithmetic(4, ev, UtW, Uty, UtX, null[0], null[1])
    rel = np.abs(var["lambda_remle"] - ref["lambda_remle"]) / np.abs(ref["lambda_remle"])
    assert np.nanmax(rel) > 1e-5 and np.nanmean(rel > 1e-6) > 0.002, "the two arithmetics agree: this test would not test anything"
    parity._cmp_stats(var, ref, 4, "two CPU arithmetics (BXD)", parity._problem(U, ev, UtW, Uty, X))


@pytest.mark.parametrize("eps,caught", [(3e-4, True), (1e-3, True), (3e-7, False)])
def test_wrong_lambda_is_rejected_on_a_well_conditioned_problem(parity, oracle, eps, caught, monkeypatch):
    """n = 500 synthetic: lambda-hat of a handful of SNPs moved by eps.  3e-4 and 1e-3 must fail the comparison (the old blanket
    bound accepted everything below 1e-3); 3e-7 is inside the first tier."""
    monkeypatch.setattr(parity, "_record", lambda line: None)
    X, U, ev, UtW, Uty, tr = parity._synthetic(oracle, 500, 300, 1, seed=600)
    l_mle, logl0 = oracle.calc_lambda_null("L", ev, UtW, Uty)
    ref = oracle.lmm_analyze(4, U, ev, UtW, Uty, X, l_mle_null=l_mle, logl_mle_H0=logl0)
    got = ref.copy()
    lam = ref["lambda_remle"]
    idx = np.flatnonzero(np.isfinite(lam) & (lam > 2e-5) & (lam < 5e4))[:4]
    got["lambda_remle"][idx] *= 1.0 + eps
    if caught:
        with pytest.raises(AssertionError, match="neither within 1e-6 nor a flipped trip count"):
            parity._cmp_stats(got, ref, 4, "perturbed %g" % eps, parity._problem(U, ev, UtW, Uty, X))
    else:
        parity._cmp_stats(got, ref, 4, "perturbed %g" % eps, parity._problem(U, ev, UtW, Uty, X))
