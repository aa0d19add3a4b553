"""Times gemma_hip_eigh on a random kinship-like matrix (n = 20000 unless given); GEMMA_HIP_EIGH_TIMING=1 prints stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
api.init(0)
g = torch.Generator(device="cuda").manual_seed(3)
# default: rank n / 2 (half of the spectrum is zero: the divide & conquer deflates heavily); "kin" as 2nd argument: full rank with a
# Marchenko-Pastur bulk, like a kinship of 2 n SNPs (next to no deflation: the merges are full-size products)
kin = len(sys.argv) > 2 and sys.argv[2] == "kin"
X = torch.randn((n, 2 * n if kin else n // 2), dtype=torch.float64, device="cuda", generator=g)
A = X @ X.T / X.shape[1]
del X
A = (A + A.T) / 2
check = os.environ.get("EIGH_PROBE_CHECK", "") == "1"  # residual / orthogonality at ANY n through the library's own GEMM
A0 = A.clone() if (n <= 8192 or check) else None
torch.cuda.empty_cache()
U = torch.empty_like(A)
w = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
t0 = time.time()
api.EigenDecomp_Zeroed(A, U, w)
torch.cuda.synchronize()
dt = time.time() - t0
msg = "eigh n=%d%s: %.2f s" % (n, " (kin)" if kin else "", dt)
if A0 is not None and n > 8192:
    eps = 2.0 ** -52
    R = torch.empty_like(A0)
    api.fast_dgemm("N", "N", 1.0, A0, U, 0.0, R)
    R.sub_(U * w[None, :])
    res = float(torch.linalg.matrix_norm(R)) / (n * eps * float(w.abs().max()))
    api.fast_dgemm("T", "N", 1.0, U, U, 0.0, R)
    R.diagonal().sub_(1.0)
    msg += ", resid %.4f orth %.4f (n*eps)" % (res, float(torch.linalg.matrix_norm(R)) / (n * eps))
elif A0 is not None:
    nrm = torch.linalg.matrix_norm(A0, 2)
    res = torch.linalg.matrix_norm(A0 @ U - U * w[None, :]) / (nrm * n * 2.2e-16)
    orth = torch.linalg.matrix_norm(U.T @ U - torch.eye(n, dtype=torch.float64, device="cuda")) / (n * 2.2e-16)
    msg += ", resid %.2f orth %.2f (n*eps)" % (float(res), float(orth))
print(msg)
