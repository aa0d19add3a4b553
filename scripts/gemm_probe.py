"""Times the UtX-shaped GEMM alone (n = B = 20000 unless given) -- used for A/B and ablation runs.
Prints the GEMMA_HIP_GEMM_* switches in force and a sampled check against torch's fp64 matmul."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randint(0, 3, (B, n), device="cuda", generator=g).to(torch.float64)
U = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g) / n ** 0.5
C = torch.empty((B, n), dtype=torch.float64, device="cuda")
api.fast_dgemm("N", "N", 1.0, X, U, 0.0, C)
torch.cuda.synchronize()
rows = torch.tensor(sorted({0, 1, 127, 128, B // 2 + 3, B - 129, B - 1}), device="cuda")
ref = X[rows] @ U
err = float((C[rows] - ref).abs().max() / ref.abs().max())
api.profile_enable(True); api.profile_read(L.STAGE_UTX_GEMM, reset=True)
for _ in range(reps):
    api.fast_dgemm("N", "N", 1.0, X, U, 0.0, C)
torch.cuda.synchronize()
ms, k = api.profile_read(L.STAGE_UTX_GEMM)
sw = " ".join("%s=%s" % (k_[15:], v) for k_, v in sorted(os.environ.items()) if k_.startswith("GEMMA_HIP_GEMM_"))
print("[%s] n=%d B=%d: %.2f ms/launch, %.2f TFLOP/s, max rel err vs torch %.1e" % (
    sw or "defaults", n, B, ms / k, 2.0 * B * n * n / (ms / k * 1e-3) / 1e12, err))
