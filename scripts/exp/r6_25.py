"""Lease 25: the stage-1 look-ahead leaves the decomposition bit-identical (GEMMA_HIP_EIGH_LOOKAHEAD=0 against the default)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gemma_amd import api
api.init(0)
for n in (8200, 9001, 14080):
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn((n, 2 * n), dtype=torch.float64, device="cuda", generator=g)
    A0 = X @ X.T / X.shape[1]
    del X
    A0 = (A0 + A0.T) / 2
    h = {}
    for la in ("1", "0"):  # 1 = at every panel, 0 = never (the default looks ahead from 20 000 rows up)
        os.environ["GEMMA_HIP_EIGH_LOOKAHEAD"] = la
        A = A0.clone()
        U = torch.empty_like(A)
        w = torch.empty(n, dtype=torch.float64, device="cuda")
        api.EigenDecomp_Zeroed(A, U, w)
        torch.cuda.synchronize()
        h[la] = (hashlib.sha256(U.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:16])
    print("n = %d: look-ahead %s, off %s -> %s" % (n, h["1"], h["0"], "IDENTICAL" if h["1"] == h["0"] else "DIFFERENT"))
