#!/bin/bash
for cfg in "4 4" "4 2" "4 8" "4 6" "8 8" "4 4"; do set -- $cfg; GEMMA_HIP_GEMM_WAVES=$1 GEMMA_HIP_GEMM_GM=$2 timeout 120 python scripts/gemm_probe.py 2>&1 | tail -1 | sed "s/^/gm=$2 /"; done
