"""Times the per-SNP stage alone on a device-resident random UtX (n = B = 20000 unless given) -- with the bracket polish
from Chebyshev series (default) and with every evaluation streaming (GEMMA_HIP_ASSOC_CHEB=0) -- and compares the two.
usage: assoc_probe.py [n] [B] [a_mode] [variants, e.g. 44,43,42]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["44"]
api.init(0)
g = torch.Generator(device="cuda").manual_seed(1)
ev = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 3
UtW = torch.randn((n, 1), dtype=torch.float64, device="cuda", generator=g)
beta = 0.7
Uty = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * (beta * ev + 1).sqrt())
UtX = torch.randn((B, n), dtype=torch.float64, device="cuda", generator=g)
Ufull = torch.empty((8, 8), dtype=torch.float64, device="cuda")  # borrowed pointer only, not read by assoc
res = {}
for cheb in ("0", "1"):
    for var in variants:
        os.environ["GEMMA_HIP_ASSOC_CHEB"] = cheb
        os.environ["GEMMA_HIP_ASSOC_VARIANT"] = var
        lmm = api.LMM(a_mode=mode, l_mle_null=beta, logl_mle_H0=-1.0)
        lmm.setup(Ufull, ev, UtW, Uty)
        out = lmm.assoc(UtX)
        torch.cuda.synchronize()
        api.profile_enable(True); api.profile_read(L.STAGE_ASSOC, reset=True)
        for _ in range(3):
            lmm.assoc(UtX, out=out)
        torch.cuda.synchronize()
        ms, k = api.profile_read(L.STAGE_ASSOC)
        print("[cheb=%s variant=%s mode=%d] n=%d B=%d: per-SNP stage %.3f ms/batch, NaN logl %d" % (
            cheb, var, mode, n, B, ms / 3, int(torch.isnan(out[:, 7]).sum())), flush=True)
        res[cheb] = out.clone()
        lmm.finish()
a, b = res["0"], res["1"]
names = ["beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score", "logl_H1"]
for j, k in enumerate(names):
    ok = torch.isfinite(a[:, j]) & torch.isfinite(b[:, j]) & (a[:, j] != 0)
    if ok.sum() == 0:
        continue
    rel = ((a[ok, j] - b[ok, j]).abs() / a[ok, j].abs())
    print("  tables vs streaming %-12s max rel %.3e, frac <= 1e-6 %.5f, NaN pattern equal %s" % (
        k, float(rel.max()), float((rel <= 1e-6).double().mean()), bool((torch.isnan(a[:, j]) == torch.isnan(b[:, j])).all())))
