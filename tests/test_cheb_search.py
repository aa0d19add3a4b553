"""The table-driven lambda search on the CPU: gemma_amd/csrc/lmm_search.hip.h (the code the per-SNP kernel runs --
GSL's Brent / Newton restated over an evaluator, Chebyshev-in-log(lambda) series of the row-0 sums, the derivative form
of LogRL_dev1/dev2 and LogL_dev1/dev2) compiled with g++ into tests/cpp/cheb_search_check and checked against

  * derivatives computed from exact n-term sums with the reference's formulas (src/lmm.cpp:866-943, :1035-1125, :544-640,
    :719-797) at random lambdas of every grid interval, and
  * the oracle's lambda-hat (REML and ML) on seeded synthetic SNPs: same bar as the GPU parity tests
    (>= 98 % within 1e-6, all within 1e-3; src/lmm.cpp:2096 reports the penultimate Newton iterate).

The series here are prepared with numpy exactly as lmm_grid.hip.h prepares them on the device (same nodes, same fit)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHEB_N = 24
MARGIN = 0.15  # of an interval's length, either side (lmm_grid.hip.h: CHEB_MARGIN)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cheb") / "cheb_search_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "cheb_search_check.cpp")])
    return exe


def _grid(l_min=1e-5, l_max=1e5, n_region=10):
    step = np.log(l_max / l_min) / n_region
    return np.array([l_min * np.exp(step * i) for i in range(n_region + 1)])


def _sums(lam, d, V):
    H = 1.0 / (lam * d + 1.0)
    return [(V * (H ** k)[:, None]).T @ V for k in (1, 2, 3)], H


def _derivs_exact(lam, d, V, c, reml):
    """(dev1, dev2) with the reference's formulas from exact sums; V columns = [w_1..w_c, x, y]."""
    n, m = V.shape[0], c + 2
    (P, PP, PPP), H = _sums(lam, d, V)
    tr1, tr2 = H.sum(), (H * H).sum()
    trace_P, trace_PP = tr1, tr2
    for p in range(c + 1):
        ww, ww2, ww3 = P[p, p], PP[p, p], PPP[p, p]
        trace_P -= ww2 / ww
        trace_PP += ww2 * ww2 / (ww * ww) - 2.0 * ww3 / ww
        Pn, PPn, PPPn = P.copy(), PP.copy(), PPP.copy()
        for a in range(m):
            for b in range(m):
                aw, bw = P[a, p], P[b, p]
                Pn[a, b] = P[a, b] - aw * bw / ww
                PPn[a, b] = PP[a, b] + aw * bw * ww2 / (ww * ww) - (aw * PP[b, p] + bw * PP[a, p]) / ww
                PPPn[a, b] = (PPP[a, b] - aw * bw * ww2 * ww2 / ww ** 3
                              - (aw * PPP[b, p] + bw * PPP[a, p] + PP[a, p] * PP[b, p]) / ww
                              + (aw * PP[b, p] * ww2 + bw * PP[a, p] * ww2 + aw * bw * ww3) / (ww * ww))
        P, PP, PPP = Pn, PPn, PPPn
    Pyy, PPyy, PPPyy = P[m - 1, m - 1], PP[m - 1, m - 1], PPP[m - 1, m - 1]
    yPKPy = (Pyy - PPyy) / lam
    yPKPKPy = (Pyy + PPPyy - 2.0 * PPyy) / (lam * lam)
    if reml:
        df = n - c - 1
        trace_PK = (df - trace_P) / lam
        trace_PKPK = (df + trace_PP - 2.0 * trace_P) / (lam * lam)
        return (-0.5 * trace_PK + 0.5 * df * yPKPy / Pyy,
                0.5 * trace_PKPK - 0.5 * df * (2.0 * yPKPKPy * Pyy - yPKPy * yPKPy) / (Pyy * Pyy),
                0.5 * abs(trace_PK) + 0.5 * abs(df * yPKPy / Pyy))
    trace_HiK = (n - tr1) / lam
    trace_HiKHiK = (n + tr2 - 2 * tr1) / (lam * lam)
    return (-0.5 * trace_HiK + 0.5 * n * yPKPy / Pyy,
            0.5 * trace_HiKHiK - 0.5 * n * (2.0 * yPKPKPy * Pyy - yPKPy * yPKPy) / (Pyy * Pyy),
            0.5 * abs(trace_HiK) + 0.5 * abs(n * yPKPy / Pyy))


def _fit_matrix():
    k = np.arange(CHEB_N)[:, None]
    m = np.arange(CHEB_N)[None, :]
    D = np.cos(np.pi * k * (m + 0.5) / CHEB_N) * 2.0 / CHEB_N
    D[0] *= 0.5
    return D  # coefficients = D @ node values


def _interval(lam_lo, lam_hi):
    a, b = np.log(lam_lo), np.log(lam_hi)
    mid, half = 0.5 * (a + b), 0.5 * (b - a) * (1.0 + 2.0 * MARGIN)
    nodes = mid + half * np.cos(np.pi * (np.arange(CHEB_N) + 0.5) / CHEB_N)
    return mid, half, nodes


QFORM_BELOW = 1e-3  # lmm_grid.hip.h: CHEB_MIN_LAMBDA -- intervals ending at or below it are tabulated in Q form


def _series(lam_lo, lam_hi, d, W, y, X):
    """-> mid, half, head (qform, s0x, s0f), fix row, per-SNP rows (l x (c + 2) CHEB_N) for the interval.  Q form (interval at or
    below lambda = 1e-3): the series are those of sum a b delta H, the constants sum a b travel beside them."""
    mid, half, nodes = _interval(lam_lo, lam_hi)
    qform = lam_hi <= QFORM_BELOW * (1 + 1e-9)
    Hn = 1.0 / (np.exp(nodes)[:, None] * d[None, :] + 1.0)      # N x n
    Ck = _fit_matrix() @ (Hn * d[None, :] if qform else Hn)       # N x n: c_k(delta_i)
    c = W.shape[1]
    F = np.column_stack([W, y])                                   # fixed variables
    fix, s0f = [], []
    for a in range(c + 1):
        for b in range(a, c + 1):
            fix.append(Ck @ (F[:, a] * F[:, b]))
            s0f.append(float(F[:, a] @ F[:, b]))
    fix.append(_fit_matrix() @ (1.0 - Hn).sum(axis=1))            # g(t) = sum (1 - H)
    fix.append(np.zeros(CHEB_N))                                  # log|H|: the final pass's, not the search's
    fix.append(_fit_matrix() @ ((1.0 - Hn) ** 2).sum(axis=1))     # sum (1 - H)^2
    rows = [(X * X) @ Ck.T]
    s0x = [(X * X).sum(axis=1)]
    for a in range(c + 1):
        rows.append((X * F[:, a][None, :]) @ Ck.T)
        s0x.append(X @ F[:, a])
    head = [np.concatenate([[1.0 if qform else 0.0], np.array([v[s] for v in s0x]), s0f]) for s in range(X.shape[0])]
    return mid, half, head, np.concatenate(fix), np.hstack(rows)


def _run(harness, tmp_path, c, reml, n, cases):
    arr = np.concatenate([np.array([c, 1.0 if reml else 0.0, n, len(cases)])] + [np.concatenate(k) for k in cases])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    arr.astype(np.float64).tofile(fin)
    subprocess.check_call([harness, str(fin), str(fout)])
    return np.fromfile(fout).reshape(len(cases), 5)


def _case(oracle, n, p, c, seed, gs=0.9, causal=(0.3, -0.2, 0.25)):
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.05, 0.5, size=p + 2 * n)
    G = rng.binomial(2, maf[:, None], size=(p + 2 * n, n)).astype(np.float64)
    K = oracle.calc_kin(G[p:], 1)
    U, ev, _ = oracle.eigen_decomp_zeroed(oracle.center_matrix(K))
    W = np.ones((n, 1)) if c == 1 else np.hstack([rng.standard_normal((n, c - 1)), np.ones((n, 1))])
    g = U @ (np.sqrt(np.maximum(ev, 0)) * rng.standard_normal(n))
    y = G[:3].T @ np.array(causal) + gs * g / g.std() + rng.standard_normal(n)
    return G[:p] @ U, ev, U.T @ W, U.T @ y


@pytest.mark.parametrize("c,reml", [(1, True), (1, False), (2, True), (3, False), (4, True)])
def test_series_derivatives_match_exact_sums(harness, tmp_path, oracle, c, reml):
    n, p = 400, 12
    UtX, d, UtW, Uty = _case(oracle, n, p, c, seed=10 + c)
    lam = _grid()
    rng = np.random.default_rng(c)
    cases, probes = [], []
    for j in range(10):
        mid, half, head, fix, rows = _series(lam[j], lam[j + 1], d, UtW, Uty, UtX)
        for s in range(p):
            pl = float(np.exp(rng.uniform(np.log(lam[j]) - 0.1, np.log(lam[j + 1]) + 0.1)))
            cases.append([np.array([lam[j], lam[j + 1], 1.0, -1.0, 1e-5, 1e5, mid, 1.0 / half, pl]), head[s], rows[s], fix])
            probes.append((s, j, pl))
    out = _run(harness, tmp_path, c, reml, n, cases)
    worst1 = worst2 = 0.0
    for (s, j, pl), o in zip(probes, out):
        assert o[2] == 1.0
        V = np.column_stack([UtW, UtX[s], Uty])
        # the reference's own formulas lose ~1e-16 / lambda in double precision (P_yy - PP_yy at small lambda): the exact
        # values are taken in long double, and the series must meet them WITHOUT a small-lambda allowance -- the intervals
        # below 1e-3 are in Q form exactly for that
        e1, e2, scale = (float(v) for v in _derivs_exact(np.longdouble(pl), d.astype(np.longdouble), V.astype(np.longdouble), c, reml))
        worst1 = max(worst1, abs(o[3] - e1) / scale)
        worst2 = max(worst2, abs(o[4] - e2) / (abs(e2) + scale / pl))
    print("series derivatives c=%d reml=%s: dev1 %.2e of its scale, dev2 %.2e" % (c, reml, worst1, worst2))
    assert worst1 < 2e-10, worst1
    assert worst2 < 2e-7, worst2


def _case_low_lambda(n, p, c, lam0, seed):
    """Rotated inputs drawn directly: a kinship spectrum with a few very large eigenvalues (population structure: delta up to
    3e4) and a phenotype with variance lam0 * delta_i + 1 in the eigenbasis, so that lambda-hat sits near lam0 << 1e-3 -- the
    decades whose intervals are tabulated in Q form."""
    rng = np.random.default_rng(seed)
    d = np.concatenate([10.0 ** rng.uniform(1.0, 4.5, size=60), rng.uniform(0.0, 2.0, size=n - 61), [0.0]])
    d = np.sort(d)
    UtW = rng.standard_normal((n, c))
    Uty = np.sqrt(lam0 * d + 1.0) * rng.standard_normal(n)
    UtX = rng.standard_normal((p, n)) * np.sqrt(0.3 * d + 1.0)[None, :]  # SNPs correlated with the structure, as real ones are
    return UtX, d, UtW, Uty


@pytest.mark.parametrize("c,gs", [(1, 0.9), (3, 0.9), (1, -3e-4), (2, -6e-5)])
def test_table_search_reproduces_the_oracles_lambda(harness, tmp_path, oracle, c, gs):
    """gs > 0: a heritable trait on a simulated kinship, lambda-hat around 1; gs < 0: -gs is the true lambda of a trait with next
    to no heritability on a spectrum with large eigenvalues, lambda-hat in the decades below 1e-3 (Q-form intervals)."""
    n, p = 500, 420
    if gs > 0:
        UtX, d, UtW, Uty = _case(oracle, n, p, c, seed=77 + c, gs=gs)
    else:
        UtX, d, UtW, Uty = _case_low_lambda(n, p, c, -gs, seed=177 + c)
    l_mle, logl0 = oracle.calc_lambda_null("L", d, UtW, Uty)
    ref = oracle.lmm_batch_UtX(4, d, UtW, Uty, np.ascontiguousarray(UtX), l_mle_null=l_mle, logl_mle_H0=logl0)
    lam = _grid()
    series = [_series(lam[j], lam[j + 1], d, UtW, Uty, UtX) for j in range(10)]
    for reml, key in ((True, "lambda_remle"), (False, "lambda_mle")):
        cases, who = [], []
        for s in range(p):
            V = np.column_stack([UtW, UtX[s], Uty])
            dv = np.array([_derivs_exact(l, d, V, c, reml)[0] for l in lam])
            lh = ref[key][s]
            br = [j for j in range(10) if dv[j] * dv[j + 1] <= 0 and lam[j] <= lh <= lam[j + 1]]
            if len(br) != 1 or not (lam[0] < lh < lam[-1]):
                continue  # boundary optimum: no bracket produced lambda-hat (the choice between candidates is not under test)
            j = br[0]
            mid, half, head, fix, rows = series[j]
            cases.append([np.array([lam[j], lam[j + 1], dv[j], dv[j + 1], 1e-5, 1e5, mid, 1.0 / half, lam[j]]), head[s], rows[s], fix])
            who.append(s)
        low = sum(1 for k in cases if k[0][1] <= QFORM_BELOW * (1 + 1e-9))
        assert len(cases) > 0.6 * p and (gs > 0 or low > 0.5 * p), (len(cases), low)
        out = _run(harness, tmp_path, c, reml, n, cases)
        ok = out[:, 0] == 0  # PB_OK; PB_OUTSIDE (3) = Newton left the widened interval: the kernel repeats those streaming
        assert np.mean(ok) > 0.97 and set(out[~ok, 0]) <= {3.0}, (np.mean(ok), set(out[:, 0]))
        rel = np.abs(out[ok, 1] - ref[key][np.array(who)[ok]]) / ref[key][np.array(who)[ok]]
        print("table search %s c=%d gs=%g: %d SNPs (%d in Q-form intervals), outside %d, max rel %.3e, frac <= 1e-6 %.4f" % (
            key, c, gs, ok.sum(), low, (~ok).sum(), rel.max(), np.mean(rel <= 1e-6)))
        assert rel.max() < 1e-3 and np.mean(rel <= 1e-6) >= 0.98
