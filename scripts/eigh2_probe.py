"""Stage-by-stage check of the two-stage eigensolver (gemma_amd/csrc/eigh2.hip.h) on the GPU, then timing.

    python scripts/eigh2_probe.py [n_small ...] [--time N]

For every small n: gemma_hip_dbg_eigh2 (band after dense -> band, tridiagonal after the bulge chase) against numpy's
eigenvalues of the input, then the full solver with GEMMA_HIP_EIGH_STAGES=2 (residual, orthogonality) beside the
one-stage path.  --time N: both paths at size N with GEMMA_HIP_EIGH_TIMING=1.
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.linalg import eigvalsh_tridiagonal

from gemma_amd import _lib as L
from gemma_amd import api


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def kin_like(n, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, n // 2))
    A = X @ X.T / (n // 2)
    A -= A.mean(0, keepdims=True)
    A -= A.mean(1, keepdims=True)
    return (A + A.T) / 2


def stages(n):
    A = kin_like(n, n)
    w_ref = np.linalg.eigvalsh(A)
    scale = np.abs(w_ref).max() * n * 2.2e-16
    band = np.zeros((n, 129))
    d = np.zeros(n)
    e = np.zeros(n - 1)
    L.check(L.lib().gemma_hip_dbg_eigh2(_p(np.ascontiguousarray(A)), n, _p(band), _p(d), _p(e)), "dbg_eigh2")
    B = np.zeros((n, n))
    for t in range(129):
        idx = np.arange(n - t)
        B[idx + t, idx] = band[: n - t, t]
        B[idx, idx + t] = band[: n - t, t]
    wb = np.linalg.eigvalsh(B)
    wt = eigvalsh_tridiagonal(d, e)
    print("n=%d  band eigenvalues vs input: %.2f n*eps   tridiagonal vs input: %.2f n*eps   (finite: %s %s)"
          % (n, np.abs(wb - w_ref).max() / scale, np.abs(wt - w_ref).max() / scale, np.isfinite(band).all(),
             np.isfinite(d).all() and np.isfinite(e).all()), flush=True)


def full(n, stages_env, check=True):
    os.environ["GEMMA_HIP_EIGH_STAGES"] = stages_env
    A = torch.from_numpy(kin_like(n, n + 1)).cuda() if n <= 8192 else None
    if A is None:
        g = torch.Generator(device="cuda").manual_seed(3)
        X = torch.randn((n, n // 2), dtype=torch.float64, device="cuda", generator=g)
        A = X @ X.T / (n // 2)
        del X
        A = (A + A.T) / 2
    A0 = A.clone() if check and n <= 8192 else None
    U = torch.empty_like(A)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    api.EigenDecomp_Zeroed(A, U, w)
    torch.cuda.synchronize()
    msg = "eigh n=%d stages=%s: %.3f s" % (n, stages_env, time.time() - t0)
    if A0 is not None:
        nrm = torch.linalg.matrix_norm(A0, 2)
        res = torch.linalg.matrix_norm(A0 @ U - U * w[None, :]) / (nrm * n * 2.2e-16)
        orth = torch.linalg.matrix_norm(U.T @ U - torch.eye(n, dtype=torch.float64, device="cuda")) / (n * 2.2e-16)
        wl = torch.linalg.eigvalsh(A0)
        big = wl > 1e-8  # EigenDecomp_Zeroed clears what is below 1e-10
        wd = (w - wl)[big].abs().max() / (nrm * n * 2.2e-16)
        msg += ", resid %.2f orth %.2f eigenvalues %.2f (n*eps)" % (float(res), float(orth), float(wd))
    print(msg, flush=True)


def main():
    args = sys.argv[1:]
    tn = None
    if "--time" in args:
        i = args.index("--time")
        tn = int(args[i + 1])
        args = args[:i] + args[i + 2:]
    api.init(0)
    for n in [int(a) for a in args]:
        stages(n)
        full(n, "2")
        full(n, "1")
    if tn:
        os.environ["GEMMA_HIP_EIGH_TIMING"] = "1"
        full(tn, "2", check=False)
        full(tn, "2", check=False)
        full(tn, "1", check=False)


if __name__ == "__main__":
    main()
