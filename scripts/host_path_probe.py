"""PCIe-inclusive rate of the host-buffer entry point (gemma_hip_lmm_batch) next to the device-resident one, n = B = 20000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gemma_amd import api, _lib as L
n, B = 20000, 20000
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
U = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g) / n ** 0.5
ev = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2
UtW = torch.randn((n, 1), dtype=torch.float64, device="cuda", generator=g)
Uty = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
nb = (n + 3) // 4
codes = torch.randint(0, 100, (B, nb * 4), device="cuda", generator=g)
codes = torch.where(codes < 1, 1, torch.where(codes < 30, 0, torch.where(codes < 70, 2, 3))).to(torch.uint8)
raw_d = (codes[:, 0::4] | (codes[:, 1::4] << 2) | (codes[:, 2::4] << 4) | (codes[:, 3::4] << 6)).contiguous()
raw_h = raw_d.cpu().numpy()
lmm = api.LMM(a_mode=1)
lmm.setup(U, ev, UtW, Uty, plink=True)
out_d = torch.empty((B, 8), dtype=torch.float64, device="cuda")
lmm.batch(raw_d, L.GENO_PLINK_2BIT, out=out_d); torch.cuda.synchronize()
lmm.batch(raw_h, L.GENO_PLINK_2BIT)
for name, fn in (("device-resident 2-bit block", lambda: (lmm.batch(raw_d, L.GENO_PLINK_2BIT, out=out_d), torch.cuda.synchronize())),
                 ("host 2-bit block (H2D 100 MB + D2H 1.3 MB)", lambda: lmm.batch(raw_h, L.GENO_PLINK_2BIT))):
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t0) / 3
    print("%s: %.1f ms per 20000-SNP block = %.0f SNPs/s" % (name, dt * 1e3, B / dt))
lmm.finish()
