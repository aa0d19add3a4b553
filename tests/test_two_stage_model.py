"""The numpy model of the two-stage tridiagonalisation (scripts/two_stage_model.py) pins, on the CPU, the three facts the
HIP implementation in gemma_amd/csrc/eigh2.hip.h relies on: (1) the tasks of the bulge chase with 2 j + k = t touch disjoint
data, so one time step may run them in any order (the persistent kernel's wait conditions are exactly this schedule);
(2) the grouped back-transformation is valid with k ascending outside and the sweep blocks descending inside (what lets
the window of Z^T slide by nb columns per group), bit-identical to the sweep-block-outer order; (3) the whole chain
A = Q1 Q2 T Q2^T Q1^T reproduces A's eigenpairs.  The GPU tests of the kernels themselves are in tests/test_gpu_eigh.py."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("two_stage_model", os.path.join(ROOT, "scripts", "two_stage_model.py"))
tsm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(tsm)


def _kin_like(n, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, n // 2))  # rank-deficient, like a kinship matrix with p < n
    A = X @ X.T / n
    A -= A.mean(0, keepdims=True)
    A -= A.mean(1, keepdims=True)
    return (A + A.T) / 2


@pytest.mark.parametrize("n,b,nb", [(97, 8, 8), (131, 16, 4), (150, 8, 4)])
def test_two_stage_model(n, b, nb):
    A = _kin_like(n, n)
    B, refl = tsm.stage1(A, b)
    assert np.abs(np.tril(B, -(b + 1))).max() < 1e-14
    assert np.abs(np.linalg.eigvalsh(B) - np.linalg.eigvalsh(A)).max() < 1e-13
    d1, e1, _, _ = tsm.band_to_tridiag(B, b, "serial")
    d, e, V2, Atri = tsm.band_to_tridiag(B, b, "pipelined")
    d2, e2, _, _ = tsm.band_to_tridiag(B, b, "pipelined_reversed")
    assert np.array_equal(d, d1) and np.array_equal(e, e1) and np.array_equal(d, d2) and np.array_equal(e, e2)
    assert np.abs(np.tril(Atri, -2)).max() < 1e-14
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w, Z = np.linalg.eigh(T)
    Uk = tsm.apply_q1(tsm.apply_q2_grouped(Z.copy(), V2, n, b, nb, "k_outer"), refl)
    Uj = tsm.apply_q1(tsm.apply_q2_grouped(Z.copy(), V2, n, b, nb, "J_outer"), refl)
    assert np.array_equal(Uk, Uj)
    assert np.abs(A @ Uk - Uk * w).max() < 1e-13 * n and np.abs(Uk.T @ Uk - np.eye(n)).max() < 1e-13 * n
