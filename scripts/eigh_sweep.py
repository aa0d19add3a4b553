"""Residual / orthogonality of gemma_hip_eigh (default path) over a sweep of sizes around the switches of round 4: the one-stage /
two-stage threshold (8000), odd sizes (the n + 1 embedding), the limits of the one-launch panel kernel (128 columns x CUs) and of
the one-position-per-workgroup chase (positions <= CUs), and the K-slice plans of every panel width in between.
    python scripts/eigh_sweep.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api
api.init(0)
EPS = 2.220446049250313e-16
sizes = [int(a) for a in sys.argv[1:]] or [7998, 8000, 8001, 9999, 12346, 16384, 20001, 32768, 32770, 33000]
worst = 0.0
for n in sizes:
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
    A = X @ X.T / X.shape[1]
    del X
    A = (A + A.T) / 2
    U = torch.empty_like(A)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    api.EigenDecomp_Zeroed(A.clone(), U, w)
    torch.cuda.synchronize()
    dt = time.time() - t0
    nrm = float(w.abs().max())
    R = A @ U
    R -= U * w[None, :]
    res = float(torch.linalg.matrix_norm(R)) / (nrm * n * EPS)
    R = U.T @ U
    R.diagonal().sub_(1.0)
    orth = float(torch.linalg.matrix_norm(R)) / (n * EPS)
    asc = bool((w[1:] >= w[:-1]).all())
    tr = abs(float(w.sum()) - float(A.diagonal().sum())) / (nrm * n * EPS)
    worst = max(worst, res, orth)
    print("n=%6d  %.2f s  resid %.3f  orth %.3f  trace %.3f (units of n eps)  ascending %s" % (n, dt, res, orth, tr, asc), flush=True)
    del A, U, w, R
    torch.cuda.empty_cache()
print("worst %.3f -> %s" % (worst, "OK" if worst < 30 else "FAIL"))
sys.exit(0 if worst < 30 else 1)
