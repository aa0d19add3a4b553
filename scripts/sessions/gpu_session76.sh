#!/bin/bash
# closing evidence of the round: the whole GPU suite (180 tests with the file-driven workflows: BIMBAM / PLINK / -loco /
# multivariate / -lm)
mkdir -p gpurun_out
( time timeout 200 python -m pytest tests -m gpu -q -x --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -14 gpurun_out/pytest_gpu.log
