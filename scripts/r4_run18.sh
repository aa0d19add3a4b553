#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_workflow_files.py -x -q -k "mvlmm" --durations=5 > $OUT/files.txt 2>&1; tail -8 $OUT/files.txt
timeout 600 python -m pytest tests/test_gpu_two_rank.py tests/test_gpu_eigh.py -x -q > $OUT/eig.txt 2>&1; tail -3 $OUT/eig.txt
