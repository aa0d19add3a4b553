"""Numerical groundwork of two round-2 items of DESIGN.md section 8 (no device code involved).

(A) The algebra behind an exact integer kinship product for hard calls: for `-gk 1` the
centred, mean-imputed SNP row is c_s = g_s - mu_s o_s (g_s in {0,1,2}, 0 where the call is missing; o_s = 1 - m_s the
observed mask; mu_s the mean over the observed calls, src/gemma_io.cpp:1511-1531), so that

    p K = sum_s c_s c_s^T = G^T G - (A + A^T) + C,
    A = a 1^T - G^T D M          a = sum_s mu_s g_s,  D = diag(mu),
    C = (sum mu^2) 1 1^T - b 1^T - 1 b^T + M^T D^2 M      b = sum_s mu_s^2 m_s.

G^T G is a product of small integers (exact in int32 for a 20 000-SNP block: every entry <= 4 * 20 000), the rank-one
terms are vectors, and the terms with the missing mask M cost nnz(M) * n -- 1 % of a dense product at 1 % missingness.
This test pins the identity against the oracle's restatement of BimbamKin / PlinkKin on hard calls with missing data;
it involves no device code."""
import numpy as np


def test_centred_kinship_equals_integer_product_plus_sparse_corrections(oracle):
    rng = np.random.default_rng(2024)
    n, p = 211, 1500
    maf = rng.uniform(0.05, 0.5, p)
    g = (rng.random((p, n)) < maf[:, None]).astype(np.int64) + (rng.random((p, n)) < maf[:, None])
    miss = rng.random((p, n)) < 0.03
    X = np.where(miss, np.nan, g.astype(np.float64))
    K_ref = oracle.calc_kin(X, 1)

    G = np.where(miss, 0, g)                      # int: 0 where missing
    M = miss.astype(np.float64)
    mu = G.sum(1) / (n - miss.sum(1))             # mean over the observed calls
    GtG = G.T @ G                                 # exact integer arithmetic
    assert GtG.dtype.kind == "i" and GtG.max() <= 4 * p
    one = np.ones(n)
    a = (mu[:, None] * G).sum(0)
    A = np.outer(a, one) - G.T.astype(np.float64) @ (mu[:, None] * M)
    b = ((mu ** 2)[:, None] * M).sum(0)
    C = (mu ** 2).sum() * np.outer(one, one) - np.outer(b, one) - np.outer(one, b) + M.T @ ((mu ** 2)[:, None] * M)
    K = (GtG - (A + A.T) + C) / p
    assert np.abs(K - K_ref).max() <= 1e-13 * np.abs(K_ref).max() * 10


def test_snp_independent_likelihood_sums_interpolate_in_log_lambda():
    """(D) The sums of a likelihood pass that do not involve the SNP -- sum_i H_i^k (ab)_i for ab in {ww, wy, yy}, sum H_i,
    sum log(lambda delta_i + 1), H_i = 1 / (lambda delta_i + 1) -- are smooth in t = log(lambda): a 13-point Chebyshev
    interpolant per unit interval of t over [1e-5, 1e5] reproduces them to ~1e-13 of their scale on a kinship-like spectrum
    of 20 000 eigenvalues (some exactly zero, a few large), i.e. far inside the 1e-6 parity bar of the statistics."""
    from numpy.polynomial import chebyshev as Ch
    rng = np.random.default_rng(0)
    n = 20000
    d = np.sort(np.concatenate([rng.gamma(0.5, 0.3, n - 50), rng.uniform(5, 400, 50)]))
    d[:3] = 0.0
    y, w = rng.standard_normal(n), rng.standard_normal(n)
    funs = {"yy1": lambda l: np.sum(y * y / (1 + l * d)), "yy3": lambda l: np.sum(y * y / (1 + l * d) ** 3),
            "wy2": lambda l: np.sum(w * y / (1 + l * d) ** 2), "tr1": lambda l: np.sum(1.0 / (1 + l * d)),
            "logdet": lambda l: np.sum(np.log(1 + l * d))}
    edges = np.arange(np.floor(np.log(1e-5)), np.ceil(np.log(1e5)) + 1)
    N = 13
    x = np.cos(np.pi * (np.arange(N) + 0.5) / N)
    for name, fun in funs.items():
        worst = 0.0
        for a, b in zip(edges[:-1], edges[1:]):
            nodes = 0.5 * (a + b) + 0.5 * (b - a) * x
            c = Ch.chebfit(x, np.array([fun(np.exp(t)) for t in nodes]), N - 1)
            tt = rng.uniform(a, b, 25)
            exact = np.array([fun(np.exp(t)) for t in tt])
            approx = Ch.chebval((tt - 0.5 * (a + b)) / (0.5 * (b - a)), c)
            worst = max(worst, float(np.max(np.abs(approx - exact)) / np.abs(exact).max()))
        assert worst < 2e-12, (name, worst)


def _reml_dev1(S1, S2, tr1, lam, n, c):
    """LogRL_dev1 (src/lmm.cpp:866-943) from the row-0 sums S_k[a][b] = sum_i H_i^k a_i b_i over the variables
    [w_1..w_c, x, y] and tr1 = sum H_i: the Schur recursions of CalcPab / CalcPPab (:283-416), then the trace terms."""
    m = c + 2
    P, PP = S1.copy(), S2.copy()
    trace_P = tr1
    for p in range(c + 1):  # project out w_1..w_c, then x
        ww, ww2 = P[p, p], PP[p, p]
        trace_P -= ww2 / ww
        Pn, PPn = P.copy(), PP.copy()
        for a in range(m):
            for b in range(m):
                Pn[a, b] = P[a, b] - P[a, p] * P[b, p] / ww
                PPn[a, b] = (PP[a, b] + P[a, p] * P[b, p] * ww2 / (ww * ww)
                             - (P[a, p] * PP[b, p] + P[b, p] * PP[a, p]) / ww)
        P, PP = Pn, PPn
    df = n - c - 1
    P_yy, PP_yy = P[m - 1, m - 1], PP[m - 1, m - 1]
    yPKPy = (P_yy - PP_yy) / lam
    trace_PK = (df - trace_P) / lam
    return -0.5 * trace_PK + 0.5 * df * yPKPy / P_yy, 0.5 * abs(trace_PK) + 0.5 * abs(df * yPKPy / P_yy)


def test_reml_derivative_from_interpolated_snp_independent_sums(bxd):
    """(D) carried to the quantity the root finder consumes: on the BXD example (n = 67, c = 3) d logRL / d lambda of every
    tested SNP, with the sums over {w, y} pairs and sum H taken from 13-node Chebyshev tables in log(lambda) and only the sums
    that involve the SNP computed from the data, agrees with the all-exact evaluation to < 1e-9 of its scale for lambda in [1e-3, 1e3]; towards both ends of [1e-5, 1e5]
    the formula's own conditioning amplifies the table error (measured here), which sets the hybrid plan of DESIGN 8D."""
    from numpy.polynomial import chebyshev as Ch
    U, d, UtW, Uty = bxd["U"], bxd["eval"], bxd["UtW"], bxd["Uty"]
    X = bxd["X"][::40].astype(np.float64)
    UtX = X @ U
    n, c = UtW.shape
    V0 = np.column_stack([UtW, np.zeros(n), Uty])  # x column filled per SNP
    fixed = [i for i in range(c + 2) if i != c]    # indices of w_1..w_c and y

    def sums(lam, V):
        H = 1.0 / (lam * d + 1.0)
        return (V * H[:, None]).T @ V, (V * (H * H)[:, None]).T @ V, H.sum()

    # tables of the SNP-independent entries on unit intervals of t = log(lambda)
    N = 13
    xs = np.cos(np.pi * (np.arange(N) + 0.5) / N)
    edges = np.arange(np.floor(np.log(1e-5)), np.ceil(np.log(1e5)) + 1)
    tables = []
    for a, b in zip(edges[:-1], edges[1:]):
        vals = []
        for t in 0.5 * (a + b) + 0.5 * (b - a) * xs:
            S1, S2, tr = sums(np.exp(t), V0)
            vals.append(np.concatenate([S1[np.ix_(fixed, fixed)].ravel(), S2[np.ix_(fixed, fixed)].ravel(), [tr]]))
        tables.append(Ch.chebfit(xs, np.array(vals), N - 1))

    def interpolated(lam):
        t = np.log(lam)
        k = min(int(np.floor(t - edges[0])), len(tables) - 1)
        a, b = edges[k], edges[k + 1]
        v = Ch.chebval((t - 0.5 * (a + b)) / (0.5 * (b - a)), tables[k])
        q = len(fixed) ** 2
        return v[:q].reshape(len(fixed), -1), v[q:2 * q].reshape(len(fixed), -1), v[2 * q]

    rng = np.random.default_rng(4)
    mid, ends = 0.0, 0.0
    for s in range(UtX.shape[0]):
        V = V0.copy()
        V[:, c] = UtX[s]
        for lam in np.exp(rng.uniform(np.log(1e-5), np.log(1e5), 6)):
            S1, S2, tr = sums(lam, V)
            exact, scale = _reml_dev1(S1, S2, tr, lam, n, c)
            T1, T2, ttr = interpolated(lam)
            A1, A2 = S1.copy(), S2.copy()
            A1[np.ix_(fixed, fixed)] = T1
            A2[np.ix_(fixed, fixed)] = T2
            approx, _ = _reml_dev1(A1, A2, ttr, lam, n, c)
            err = abs(approx - exact) / scale  # relative to the size of the two terms dev1 is the difference of
            if 1e-3 <= lam <= 1e3:
                mid = max(mid, err)
            else:
                ends = max(ends, err)
    # the formula divides differences of the sums by lambda (small lambda) or multiplies them (large lambda): a table error of
    # ~1e-12 is amplified ~ 1/lambda resp. lambda towards the ends of [1e-5, 1e5].  Inside [1e-3, 1e3] -- where the root
    # finder spends its iterations -- the tables are good to 1e-9; the two outer decades either side keep the exact pass.
    assert mid < 1e-9, mid
    assert ends < 1e-6, ends
