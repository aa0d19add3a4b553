#!/bin/bash
# the round's very last GPU seconds: the library with the 16-row records kernel as its default -- the exactness tests of the int8 path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_31; mkdir -p $OUT
timeout 42 python -m pytest tests/test_gpu_parity.py -x -q -k "records_kernel or utx_int8_sparse or utx_int8_digit" > $OUT/int8_tests.txt 2>&1; echo "rc=$?" >> $OUT/int8_tests.txt
tail -4 $OUT/int8_tests.txt
