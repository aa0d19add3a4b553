"""why does test_lmm_reference_xlarge_layout_and_batching differ?  (round 6 debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from gemma_amd import api, _lib as L
from oracle import oracle as O
from test_gpu_parity import _synthetic
api.init(0)
X, U, ev, UtW, Uty, _ = _synthetic(O, 300, 200, 1, seed=77)
Xi = O.impute_mean(X)
for scale in ("max", "pow2"):
    os.environ["GEMMA_HIP_I8_SCALE"] = scale
    lmm = api.LMM(a_mode=1)
    lmm.setup(U, ev, UtW, Uty)
    Xlarge = np.zeros((300, 256)); Xlarge[:, :200] = Xi.T
    ua = lmm.dbg_utx(np.ascontiguousarray(Xlarge[:, :200]), L.GENO_F64_IDV_MAJOR, 1)
    pa = api.last_utx_path()
    ub = np.concatenate([lmm.dbg_utx(X[s:s + 64], L.GENO_F64_SNP_MAJOR, 1) for s in range(0, 200, 64)])
    pb = api.last_utx_path()
    uc = lmm.dbg_utx(X, L.GENO_F64_SNP_MAJOR, 1)
    a = lmm.batch(Xlarge[:, :200], L.GENO_F64_IDV_MAJOR)
    b = np.concatenate([lmm.batch(X[s:s + 64], L.GENO_F64_SNP_MAJOR) for s in range(0, 200, 64)])
    lmm.finish()
    d = ua != ub
    print(scale, "paths", pa, pb, "utx differing entries a-vs-b:", int(d.sum()), "rows:", np.flatnonzero(d.any(axis=1))[:10], "b-vs-c:", int((ub != uc).sum()))
    if d.any():
        r, c = np.argwhere(d)[0]
        print("  first", r, c, repr(ua[r, c]), repr(ub[r, c]), "missing in row:", int(np.isnan(X[r]).sum()), "mean", repr(np.nanmean(X[r])))
    for k in a.dtype.names:
        print("  ", k, "equal" if np.array_equal(a[k], b[k], equal_nan=True) else "DIFFERENT (%d)" % int((a[k] != b[k]).sum()))
