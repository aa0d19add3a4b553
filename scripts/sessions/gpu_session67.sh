#!/bin/bash
# first GPU run of the multivariate LMM: parity tests, then throughput at BASELINE config 5's shape
timeout 600 python -m pytest tests/test_gpu_mvlmm.py -m gpu -q -x 2>&1 | tail -15
timeout 500 python scripts/mvlmm_probe.py 10000 8192 3 1 2>&1 | grep -v amdgpu.ids | tail -8
