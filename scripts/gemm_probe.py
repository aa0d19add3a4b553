"""Times the UtX-shaped GEMM alone (n = B = 20000 unless given) -- used for A/B and ablation runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randint(0, 3, (B, n), device="cuda", generator=g).to(torch.float64)
U = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g) / n ** 0.5
C = torch.empty((B, n), dtype=torch.float64, device="cuda")
api.fast_dgemm("N", "N", 1.0, X, U, 0.0, C)
torch.cuda.synchronize()
api.profile_enable(True); api.profile_read(L.STAGE_UTX_GEMM, reset=True)
for _ in range(3):
    api.fast_dgemm("N", "N", 1.0, X, U, 0.0, C)
torch.cuda.synchronize()
ms, k = api.profile_read(L.STAGE_UTX_GEMM)
print("waves=%s ablate=%s: %.2f ms/launch, %.2f TFLOP/s" % (os.environ.get("GEMMA_HIP_GEMM_WAVES", "8"), os.environ.get("GEMMA_HIP_GEMM_ABLATE", "0"), ms / k, 2.0 * B * n * n / (ms / k * 1e-3) / 1e12))
