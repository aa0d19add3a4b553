"""Kinship of a PLINK block: the exact-integer path (default for -gk 1) against the fp64 SYRK (GEMMA_HIP_KIN_I8=0), timing
and agreement, with the round-3 switches (GEMMA_HIP_KIN_LISTS, GEMMA_HIP_KIN_UPPER) off and on; usage: kin_probe.py [n] [snps];
KIN_PROBE_ONLY=<index> runs one variant (for rocprofv3)"""
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
MODES = [("fp64 SYRK", {"GEMMA_HIP_KIN_I8": "0"}),
         ("integer, round-2 kernels (all tiles, in-block lists)", {"GEMMA_HIP_KIN_I8": "1", "GEMMA_HIP_KIN_LISTS": "0", "GEMMA_HIP_KIN_UPPER": "0"}),
         ("integer, correction on lists", {"GEMMA_HIP_KIN_I8": "1", "GEMMA_HIP_KIN_LISTS": "1", "GEMMA_HIP_KIN_UPPER": "0"}),
         ("integer, lists + upper-triangle tiles (default)", {"GEMMA_HIP_KIN_I8": "1", "GEMMA_HIP_KIN_LISTS": "1", "GEMMA_HIP_KIN_UPPER": "1"})]
only = os.environ.get("KIN_PROBE_ONLY")
for name, env in MODES:
    if only is not None and name != MODES[int(only)][0]:
        continue
    os.environ.update(env)
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
    print("[%s] n=%d p=%d: %.3f s (%.1f ms per 20000-SNP block)" % (name, n, p, dt, dt / len(blocks) * 1e3), flush=True)
    res[name] = K
ref = res.get(MODES[0][0])
for name, K in res.items():
    if ref is not None and name != MODES[0][0]:
        d = (ref - K).abs().max().item() / ref.abs().max().item()
        print("%s: max |K - K_fp64| / max |K| = %.2e; symmetric: %s" % (name, d, bool((K - K.T).abs().max().item() == 0.0)))
