#!/bin/bash
# HIP path against the reference's own outputs (tests/golden/ref_*.npz from oracle/_ref/gemma)
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_reference.py -m gpu -q --durations=8 2>&1 | tail -60 | tee gpurun_out/ref_tests.log
