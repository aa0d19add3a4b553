"""Times the multivariate per-SNP stage (gemma_hip_mvlmm_batch) on synthetic data shaped like BASELINE config 5
(n = 10000, 3 phenotypes) and, on a sample of SNPs, the CPU oracle beside it.  usage: mvlmm_probe.py [n] [p] [d] [a_mode]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
from gemma_amd import api, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
api.init(0)
rng = np.random.default_rng(10000)
m = 2 * n
maf = rng.uniform(0.05, 0.5, m)
Gk = rng.binomial(2, maf[None, :], size=(n, m)).astype(np.float64)
Gk -= Gk.mean(0)
K = Gk @ Gk.T / m
del Gk
U, ev = np.zeros((n, n)), np.zeros(n)
t0 = time.time()
api.EigenDecomp_Zeroed(K.copy(), U, ev)
print("eigen n=%d: %.1f s" % (n, time.time() - t0))
Lg = np.tril(rng.standard_normal((d, d))) * 0.25 + np.eye(d) * 1.3
Le = np.tril(rng.standard_normal((d, d))) * 0.25 + np.eye(d)
Y = (U * np.sqrt(np.maximum(ev, 0))) @ rng.standard_normal((n, d)) @ Lg.T + rng.standard_normal((n, d)) @ Le.T
G = rng.binomial(2, rng.uniform(0.05, 0.5, p)[:, None], size=(p, n)).astype(np.float64)
UtW = U.T @ np.ones((n, 1))
UtY = U.T @ Y
mv = api.MVLMM(a_mode=a_mode)
t0 = time.time()
null = mv.fit_null(ev, UtW, UtY)
print("null block: %.2f s, logl_remle %.4f logl_mle %.4f" % (time.time() - t0, null["logl_remle"], null["logl_mle"]))
lmm = api.LMM(a_mode=a_mode)
lmm.setup(U, ev, UtW, np.ascontiguousarray(UtY[:, 0]))
opt = mv._opt()
L.check(L.lib().gemma_hip_mvlmm_set(d, UtY.ctypes.data, C.byref(mv._null_struct), C.byref(opt)), "set")
v = d * (d + 1) // 2
out = np.zeros((p, d + 3 * v + 3))
L.lib().gemma_hip_profile_enable(1)
for rep in range(2):
    t0 = time.time()
    L.check(L.lib().gemma_hip_mvlmm_batch(L.GENO_F64_SNP_MAJOR, G.ctypes.data, p, n, out.ctypes.data), "batch")
    dt = time.time() - t0
ms, cnt = C.c_double(), C.c_long()
L.lib().gemma_hip_profile_read(L.STAGE_ASSOC, C.byref(ms), C.byref(cnt), 1)
per = ms.value / max(cnt.value, 1)
print("mvlmm batch: p=%d n=%d d=%d a_mode=%d: wall %.3f s, per-SNP stage %.1f ms -> %.0f SNPs/s (stage), %.0f SNPs/s (wall)"
      % (p, n, d, a_mode, dt, per, p / (per / 1e3), p / dt))
print("  p < 1e-3: %d SNPs" % int((out[:, d + 3 * v + (0 if a_mode in (1, 4) else a_mode - 1)] < 1e-3).sum()))
lmm.finish()
# CPU oracle on a sample
from oracle import oracle as O
cfg = O.mv_cfg()
ns = 16
UtWt, UtYt = np.ascontiguousarray(UtW.T), np.ascontiguousarray(UtY.T)
onull = {"Vg_mle": null["Vg_mle"], "Ve_mle": null["Ve_mle"], "B_mle": null["B_mle"], "logl_mle": null["logl_mle"]}
UtX = np.ascontiguousarray(G[:ns] @ U)
t0 = time.time()
ref = O.mvlmm_batch(a_mode, cfg, ev, UtWt, UtYt, UtX, onull)
dt = time.time() - t0
print("oracle (1 core): %d SNPs in %.2f s -> %.1f SNPs/s" % (ns, dt, ns / dt))
key = {1: "p_wald", 2: "p_lrt", 3: "p_score", 4: "p_wald"}[a_mode]
col = d + 3 * v + {1: 0, 2: 1, 3: 2, 4: 0}[a_mode]
print("  max rel diff of %s on the sample: %.2e" % (key, np.max(np.abs(out[:ns, col] - ref[key]) / ref[key])))
