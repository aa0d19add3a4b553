"""Null model (gemma_hip_lmm_null) on a kinship spectrum in other units: GPU against the oracle for several n and scales."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gemma_amd import api
from oracle import oracle as O
api.init(0)
for n in (400, 3000, 20000):
    rng = np.random.default_rng(5)
    ev = np.sort(np.concatenate([[0.0], [0.035 * n], rng.uniform(0.3, 2.2, size=n - 2)]))
    UtW = rng.standard_normal((n, 1))
    Uty = np.sqrt(3.2 * ev + 1) * rng.standard_normal(n)
    for S in (1.0, 1e2, 1e4, 1.0798e4):
        ref = O.calc_lambda_null("R", ev * S, UtW, Uty)
        nm = api.CalcLambdaNull(ev * S, UtW, Uty, trace_G=float((ev * S).mean()))
        print("n=%d S=%g oracle %.6e %.6f  gpu %.6e %.6f  mle %.6e" % (n, S, ref[0], ref[1], nm["l_remle_null"], nm["logl_remle_H0"], nm["l_mle_null"]), flush=True)
