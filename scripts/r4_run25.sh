#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_25; mkdir -p $OUT
timeout 60 scripts/abl_bin/mfma16_layout_probe > $OUT/mfma16_layout.txt 2>&1
cat $OUT/mfma16_layout.txt
