"""Lease 31: what a first-pass block costs inside gemma_hip_snp_qc (n = 20 000, 20 000 SNPs of PLINK bytes = 100 MB from pageable memory),
with the kernel trace of the calls; and a plain pageable upload of the same block for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gemma_amd import api, _lib as L
api.init(0)
n, l = 20000, 20000
rng = np.random.default_rng(1)
raw = rng.integers(0, 256, size=(l, n // 4), dtype=np.uint8)
raw2 = raw.copy()
ind = np.ones(n, dtype=np.int32)
ind[rng.random(n) < 0.017] = 0
W = np.ones((int(ind.sum()), 1))
for it in range(6):
    t0 = time.perf_counter()
    api.SnpQC(raw if it % 2 else raw2, L.GENO_PLINK_2BIT, ind, W)
    print("snp_qc call %d: %.2f ms" % (it, (time.perf_counter() - t0) * 1e3))
d = torch.empty((l, n // 4), dtype=torch.uint8, device="cuda")
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.copy_(torch.from_numpy(raw if it % 2 else raw2))
    torch.cuda.synchronize()
    print("pageable upload of 100 MB (torch copy_): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
