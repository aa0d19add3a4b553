"""The algebra behind an exact integer kinship product for hard calls: for `-gk 1` the
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
