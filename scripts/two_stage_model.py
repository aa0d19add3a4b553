"""numpy model of the two-stage tridiagonalisation in gemma_amd/csrc/eigh2.hip.h (dense -> band -> tridiagonal) and of
the two back-transformations, written to pin down the index conventions, the task schedule of the bulge chase and the
order in which the grouped reflectors may be applied.  Not a product path and not test infrastructure for parity: it
checks the ALGORITHM (A U = U diag(w), U^T U = I) at small sizes on the CPU.

    python scripts/two_stage_model.py [n] [b] [nb]
"""
import sys

import numpy as np


def house(x):
    """v (v[0] = 1), tau, beta with (I - tau v v^T) x = beta e_1 (LAPACK dlarfg convention)."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0:
        return np.concatenate([[1.0], np.zeros(len(x) - 1)]), 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def stage1(A, b):
    """dense -> band of half-width b.  Panel p covers columns [j0, j0 + b): QR of A[j0 + b:, j0:j0 + b], then the
    two-sided update of the trailing matrix.  Returns the band matrix (dense storage) and the list of (row0, V, T)."""
    A = A.copy()
    n = A.shape[0]
    refl = []
    for j0 in range(0, n - b - 1, b):
        r0 = j0 + b
        kp = min(b, n - j0)  # columns of the panel
        P = A[r0:, j0:j0 + kp]
        m = P.shape[0]
        kk = min(kp, m - 1) if m > 1 else 0
        V = np.zeros((m, kk))
        tau = np.zeros(kk)
        for c in range(kk):
            v, t, beta = house(P[c:, c].copy())
            V[c:, c] = v
            tau[c] = t
            P[c:, c:] -= t * np.outer(v, v @ P[c:, c:])
        if kk == 0:
            continue
        # forward compact WY: Q = I - V T V^T
        T = np.zeros((kk, kk))
        for i in range(kk):
            T[i, i] = tau[i]
            if i:
                T[:i, i] = -tau[i] * T[:i, :i] @ (V[:, :i].T @ V[:, i])
        A[j0:j0 + kp, r0:] = A[r0:, j0:j0 + kp].T
        # two-sided update of the trailing matrix A22 <- Q^T A22 Q as A22 - V W^T - W V^T
        A22 = A[r0:, r0:]
        Y = A22 @ V @ T                    # m x kk
        W = Y - 0.5 * V @ (T.T @ (V.T @ Y))
        A22 -= V @ W.T + W @ V.T
        refl.append((r0, V, T))
    return A, refl


def band_to_tridiag(B, b, schedule="pipelined"):
    """Bulge chase on the band matrix (dense storage, both triangles kept for the model).  Task (j, k) works on rows
    R_k = [j + 1 + k b, j + 1 + (k + 1) b).  schedule = "serial": sweep after sweep; "pipelined": every time step t runs all
    tasks with 2 j + k = t on the state left by step t - 1 (what the persistent kernel's flags enforce) -- the two must
    agree to rounding, which is the check that the tasks of one time step touch disjoint data."""
    A = B.copy()
    n = A.shape[0]
    V2 = {}

    def task(j, k, A):
        r = j + 1 + k * b
        L = min(b, n - r)
        if L <= 0:
            return
        rows = slice(r, r + L)
        if k == 0:
            x = A[rows, j].copy()
        else:
            vp, tp = V2[(j, k - 1)]
            cp = slice(r - b, r)  # rows of the previous task (always a full block)
            E = A[rows, cp]
            E -= tp * np.outer(E @ vp, vp)
            x = E[:, 0].copy()
        if L == 1 and k > 0:
            # a single row: nothing to annihilate, the reflector is the identity
            V2[(j, k)] = (np.ones(1), 0.0)
            A[cp, rows] = A[rows, cp].T
            return
        v, tau, beta = house(x)
        V2[(j, k)] = (v, tau)
        if k == 0:
            A[rows, j] = 0.0
            A[r, j] = beta
            A[j, rows] = A[rows, j]
        else:
            E[:, 0] = 0.0
            E[0, 0] = beta
            E[:, 1:] -= tau * np.outer(v, v @ E[:, 1:])
            A[cp, rows] = E.T
        D = A[rows, rows]
        p = tau * (D @ v)
        w = p - 0.5 * tau * (v @ p) * v
        D -= np.outer(v, w) + np.outer(w, v)

    nsweep = n - 2
    kmax = lambda j: -(-(n - 1 - j) // b)  # tasks of sweep j
    if schedule == "serial":
        for j in range(nsweep):
            for k in range(kmax(j)):
                task(j, k, A)
    else:
        tmax = 2 * (nsweep - 1) + kmax(0) + 2
        for t in range(tmax + 1):
            todo = [(j, t - 2 * j) for j in range(nsweep) if 0 <= t - 2 * j < kmax(j)]
            if schedule == "pipelined_reversed":  # the tasks of one step touch disjoint data: any order gives the same bits
                todo = todo[::-1]
            for (j, k) in todo:
                task(j, k, A)
    d = np.diag(A).copy()
    e = np.diag(A, -1).copy()
    return d, e, V2, A


def apply_q2_grouped(Z, V2, n, b, nb, order="k_outer"):
    """Z <- Q2 Z with Q2 = prod_{j ascending} prod_{k ascending} H_{j,k}: sweep blocks J from the last to the first, inside
    a block k ASCENDING, each group (J, k) as one compact-WY block reflector of the nb sweeps (rows shift by one per sweep)."""
    nsweep = n - 2
    kmax = lambda j: -(-(n - 1 - j) // b)
    if order == "J_outer":
        seq = [(J0, k) for J0 in reversed(range(0, nsweep, nb)) for k in range(kmax(J0))]
    else:  # what q2_apply_kernel does: k ascending outside, sweep blocks descending inside (the window slides by nb columns)
        seq = [(J0, k) for k in range(kmax(0)) for J0 in reversed(range(0, nsweep, nb)) if k < kmax(J0)]
    for (J0, k) in seq:
        js = list(range(J0, min(J0 + nb, nsweep)))
        if True:
            r0 = J0 + 1 + k * b
            width = min(b + len(js) - 1, n - r0)
            if width <= 0:
                continue
            V = np.zeros((width, len(js)))
            tau = np.zeros(len(js))
            for c, j in enumerate(js):
                if (j, k) in V2:
                    v, t = V2[(j, k)]
                    V[c:c + len(v), c] = v
                    tau[c] = t
            T = np.zeros((len(js), len(js)))
            for i in range(len(js)):
                T[i, i] = tau[i]
                if i:
                    T[:i, i] = -tau[i] * T[:i, :i] @ (V[:, :i].T @ V[:, i])
            # product of the group in sweep order: H_{J0,k} H_{J0+1,k} ... = I - V T V^T
            Zs = Z[r0:r0 + width]
            Zs -= V @ (T @ (V.T @ Zs))
    return Z


def apply_q1(Z, refl):
    for (r0, V, T) in reversed(refl):
        Zs = Z[r0:]
        Zs -= V @ (T @ (V.T @ Zs))
    return Z


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, n // 2))  # rank-deficient, like a kinship matrix with p < n
    A = X @ X.T / n
    A -= A.mean(0, keepdims=True)
    A -= A.mean(1, keepdims=True)
    B, refl = stage1(A, b)
    off = np.abs(np.tril(B, -(b + 1))).max()
    print("stage 1: max |B| outside the band %.2e" % off)
    d1, e1, V2s, _ = band_to_tridiag(B, b, "serial")
    d, e, V2, Atri = band_to_tridiag(B, b, "pipelined")
    d2, e2, _, _ = band_to_tridiag(B, b, "pipelined_reversed")
    print("stage 2: pipelined vs serial schedule: %.2e %.2e; tasks of a step in reverse order: %.2e %.2e"
          % (np.abs(d - d1).max(), np.abs(e - e1).max(), np.abs(d - d2).max(), np.abs(e - e2).max()))
    print("stage 2: max |A| outside the tridiagonal %.2e" % np.abs(np.tril(Atri, -2)).max())
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w, Z = np.linalg.eigh(T)
    U = apply_q1(apply_q2_grouped(Z.copy(), V2, n, b, nb), refl)
    U2 = apply_q1(apply_q2_grouped(Z.copy(), V2, n, b, nb, "J_outer"), refl)
    print("Q2 group order k-outer vs J-outer: %.2e" % np.abs(U - U2).max())
    scale = np.abs(A).max() * n
    print("eigenvalues vs LAPACK %.2e" % (np.abs(w - np.linalg.eigvalsh(A)).max() / np.abs(w).max()))
    print("||A U - U w|| / (n |A|) %.2e   ||U^T U - I|| %.2e" % (np.abs(A @ U - U * w).max() / scale,
                                                              np.abs(U.T @ U - np.eye(n)).max()))


if __name__ == "__main__":
    main()
