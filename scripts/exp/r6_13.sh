#!/bin/bash
# lease 12: stage-2 back-transformation, eight wavefronts per workgroup with the pack by LDS-DMA into a double buffer (GEMMA_HIP_EIGH_Q2_WAVES=8)
GEMMA_HIP_EIGH_Q2_WAVES=8 timeout 1500 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_two_rank.py -m gpu -q -x > $OUT/pytest_eigh_w8.txt 2>&1; tail -4 $OUT/pytest_eigh_w8.txt
python - <<'PY'
# bit-identity of the two kernels' results on three shapes (two-stage forced)
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from gemma_amd import api
from test_gpu_eigh import _sym
api.init(0)
os.environ["GEMMA_HIP_EIGH_STAGES"] = "2"
for n, kind, dynq in ((1538, "kinship", "1"), (2307, "random", "0"), (4100, "kinship", "1"), (700, "lowrank", "1")):
    os.environ["GEMMA_HIP_EIGH_Q2_DYNAMIC"] = dynq
    A = _sym(n, n + 1, kind)
    out = {}
    for wv in ("4", "8"):
        os.environ["GEMMA_HIP_EIGH_Q2_WAVES"] = wv
        U, w = np.zeros((n, n)), np.zeros(n)
        api.EigenDecomp_Zeroed(A.copy(), U, w)
        out[wv] = (U, w)
    print("n=%d %s dynamic=%s: U bit-identical %s, w %s" % (n, kind, dynq, np.array_equal(out["4"][0], out["8"][0]), np.array_equal(out["4"][1], out["8"][1])))
PY
{
for W in 4 8; do for n in 20000 50000; do
  echo "== n = $n (kin), Q2 waves $W"
  GEMMA_HIP_EIGH_Q2_WAVES=$W GEMMA_HIP_EIGH_TIMING=1 timeout 600 python scripts/eigh_probe.py $n kin 2>&1 | grep -E "gemma_hip_eigh n=.*two-stage|eigh n="
done; done
} > $OUT/eigh_q2_w8.txt 2>&1; cat $OUT/eigh_q2_w8.txt | cut -c1-250
