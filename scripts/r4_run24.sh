#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_24; mkdir -p $OUT
timeout 120 scripts/abl_bin/mfma_power_probe 20000 > $OUT/mfma_power.txt 2>&1
cat $OUT/mfma_power.txt
