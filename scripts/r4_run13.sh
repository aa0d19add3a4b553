#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_13; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_mvlmm.py -x -q --durations=12 > $OUT/mv.txt 2>&1
tail -30 $OUT/mv.txt
timeout 600 python -m pytest tests/test_gpu_reference.py -x -q -k "mvlmm" --durations=8 > $OUT/mvref.txt 2>&1
tail -20 $OUT/mvref.txt
