"""Times the per-SNP stage alone on a device-resident random UtX (n = B = 20000 unless given)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
api.init(0)
g = torch.Generator(device="cuda").manual_seed(1)
U = torch.zeros((8, 8), dtype=torch.float64, device="cuda")  # not used by assoc
ev = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 3
UtW = torch.randn((n, 1), dtype=torch.float64, device="cuda", generator=g)
beta = 0.7
Uty = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * (beta * ev + 1).sqrt())
UtX = torch.randn((B, n), dtype=torch.float64, device="cuda", generator=g)
lmm = api.LMM(a_mode=1)
Ufull = torch.empty((n, n), dtype=torch.float64, device="cuda")  # borrowed pointer only
lmm.setup(Ufull, ev, UtW, Uty)
out = lmm.assoc(UtX)
torch.cuda.synchronize()
api.profile_enable(True); api.profile_read(L.STAGE_ASSOC, reset=True)
for _ in range(3):
    lmm.assoc(UtX, out=out)
torch.cuda.synchronize()
ms, k = api.profile_read(L.STAGE_ASSOC)
sw = " ".join("%s=%s" % (k_[10:], v) for k_, v in sorted(os.environ.items()) if k_.startswith("GEMMA_HIP_"))
print("[%s] n=%d B=%d: per-SNP stage %.2f ms/batch, NaN p_wald %d" % (sw or "defaults", n, B, ms / 3, int(torch.isnan(out[:, 4]).sum())))
lmm.finish()
