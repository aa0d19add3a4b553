"""Times the U^T x stage of a PLINK batch alone (no eigensolver): random U, random 2-bit block on the device.
Usage: i8_probe.py [n] [B] [reps]; GEMMA_HIP_UTX_I8 / GEMMA_HIP_I8_* select the path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
api.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
U = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g) / n ** 0.5
ev = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2
UtW = torch.randn((n, 1), dtype=torch.float64, device="cuda", generator=g)
Uty = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
nb = (n + 3) // 4
# 2-bit codes with ~1 % missing (code 1)
codes = torch.randint(0, 100, (B, nb * 4), device="cuda", generator=g)
codes = torch.where(codes < 1, 1, torch.where(codes < 30, 0, torch.where(codes < 70, 2, 3))).to(torch.uint8)
mode = os.environ.get("PROBE_GENO", "")
if mode == "zeros":      # every call homozygous 0: the genotype operand is all zero
    codes.fill_(3)
elif mode == "nomiss":   # no missing calls: the mask operand is all zero
    codes = torch.where(codes == 1, 3, codes).to(torch.uint8)
if os.environ.get("PROBE_U", "") == "pow2":  # U entries exact powers of two: 6 of the 7 digits are zero
    U = torch.sign(U) * torch.exp2(torch.floor(torch.log2(U.abs())))
raw = (codes[:, 0::4] | (codes[:, 1::4] << 2) | (codes[:, 2::4] << 4) | (codes[:, 3::4] << 6)).contiguous()
lmm = api.LMM(a_mode=3)  # score test only: the per-SNP stage is one pass, the probe is about U^T x
lmm.setup(U, ev, UtW, Uty, plink=True)
out = torch.empty((B, 8), dtype=torch.float64, device="cuda")
lmm.batch(raw, L.GENO_PLINK_2BIT, out=out)
torch.cuda.synchronize()
api.profile_enable(True)
for st in (L.STAGE_INGEST, L.STAGE_UTX_GEMM, L.STAGE_ASSOC):
    api.profile_read(st, reset=True)
for _ in range(reps):
    lmm.batch(raw, L.GENO_PLINK_2BIT, out=out)
torch.cuda.synchronize()
ms, k = api.profile_read(L.STAGE_UTX_GEMM)
mi, ki = api.profile_read(L.STAGE_INGEST)
sw = " ".join("%s=%s" % (k_.replace("GEMMA_HIP_", ""), v) for k_, v in sorted(os.environ.items()) if k_.startswith("GEMMA_HIP_") or k_.startswith("PROBE_"))
mp, _ = api.profile_read(L.STAGE_UTX_POST)
api.profile_enable(False)
# agreement of the two products on the first 128 SNPs (host round trip through the debug entry point)
sub = raw[:128].cpu().numpy()
a = lmm.dbg_utx(sub, L.GENO_PLINK_2BIT, 0)
b = lmm.dbg_utx(sub, L.GENO_PLINK_2BIT, 1)
import numpy as np
rel = float(np.max(np.abs(a - b)) / np.max(np.abs(a)))
print("[%s] n=%d B=%d: U^T x %.2f ms/batch + combine %.2f ms (ingest %.2f ms); fp64 vs int8-digit max diff / max |UtX| = %.1e" % (
    sw or "defaults", n, B, ms / reps, mp / reps, mi / reps, rel))
lmm.finish()
