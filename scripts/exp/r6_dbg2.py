"""Which buffer of the eigensolver is read before it is written?  GEMMA_HIP_EIGH_POISON=1 fills every workspace block with NaN bit
patterns; the stage diagnostics and the end-to-end solve say where NaNs come out (round 6 debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import ctypes as C
import numpy as np
from gemma_amd import api, _lib as L
from test_gpu_eigh import _sym
api.init(0)
p = lambda a: C.c_void_p(a.ctypes.data)
for n, stages in ((770, "2"), (1026, "2"), (517, "1"), (770, "1")):
    os.environ["GEMMA_HIP_EIGH_STAGES"] = stages
    A = _sym(n, 5, "random")
    wr = np.linalg.eigvalsh(A)
    for poison in ("0", "1"):
        os.environ["GEMMA_HIP_EIGH_POISON"] = poison
        U, w = np.zeros((n, n)), np.zeros(n)
        try:
            api.EigenDecomp_Zeroed(A.copy(), U, w)
            msg = "eval err %.3g, orth %.3g, NaN in U %d, in w %d" % (np.nanmax(np.abs(np.sort(w) - wr)) / np.abs(wr).max(),
                                                                     np.linalg.norm(np.nan_to_num(U.T @ U) - np.eye(n)), int(np.isnan(U).sum()), int(np.isnan(w).sum()))
        except Exception as e:
            msg = "raised " + repr(e)[:200]
        print("n=%d stages=%s poison=%s: %s" % (n, stages, poison, msg))
        if stages == "2" and n % 2 == 0:
            band, d, e = np.zeros((n, 129)), np.zeros(n), np.zeros(n - 1)
            rc = L.lib().gemma_hip_dbg_eigh2(p(A.copy()), n, p(band), p(d), p(e))
            print("   dbg_eigh2 rc=%d: NaN in band %d, d %d, e %d" % (rc, int(np.isnan(band).sum()), int(np.isnan(d).sum()), int(np.isnan(e).sum())))
        dd, ee, tau, VT = np.zeros(n), np.zeros(n - 1), np.zeros(n), np.zeros((n, n))
        if stages == "1":
            rc = L.lib().gemma_hip_dbg_tridiag(p(A.copy()), n, p(dd), p(ee), p(tau), p(VT))
            print("   dbg_tridiag rc=%d: NaN in d %d e %d tau %d VT %d" % (rc, int(np.isnan(dd).sum()), int(np.isnan(ee).sum()), int(np.isnan(tau).sum()), int(np.isnan(VT).sum())))
            from scipy.linalg import eigh_tridiagonal
            d0, e0 = np.random.default_rng(1).standard_normal(n), np.random.default_rng(2).standard_normal(n - 1)
            w2, ZT = np.zeros(n), np.zeros((n, n))
            rc = L.lib().gemma_hip_dbg_stedc(p(d0), p(e0), n, p(w2), p(ZT))
            print("   dbg_stedc rc=%d: NaN in w %d ZT %d, eval err %.3g" % (rc, int(np.isnan(w2).sum()), int(np.isnan(ZT).sum()),
                                                                          np.nanmax(np.abs(w2 - eigh_tridiagonal(d0, e0, eigvals_only=True)))))
