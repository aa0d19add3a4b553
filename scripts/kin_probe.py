"""Kinship of a PLINK block: the exact-integer path (default for -gk 1) against the fp64 SYRK (GEMMA_HIP_KIN_I8=0), timing
and agreement; usage: kin_probe.py [n] [snps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gemma_amd import api, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
api.init(0)
dev = torch.device("cuda", 0)
torch.manual_seed(3)
gen = torch.Generator(device=dev).manual_seed(3)
blocks = [bench.synth_block(torch, n, min(20000, p - s), gen, dev) for s in range(0, p, 20000)]
res = {}
for mode in ("0", "1"):
    os.environ["GEMMA_HIP_KIN_I8"] = mode
    K = torch.empty((n, n), dtype=torch.float64, device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.kin_begin(n, 1)
        for b in blocks:
            api.kin_add(b, L.GENO_PLINK_2BIT)
        api.kin_end(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("[KIN_I8=%s] n=%d p=%d: %.3f s (%.1f ms per 20000-SNP block)" % (mode, n, p, dt, dt / len(blocks) * 1e3), flush=True)
    res[mode] = K
d = (res["0"] - res["1"]).abs().max().item() / res["0"].abs().max().item()
print("max |K_int - K_fp64| / max |K| = %.2e; symmetric: %s" % (d, bool((res["1"] - res["1"].T).abs().max().item() == 0.0)))
