#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_28; mkdir -p $OUT
timeout 60 scripts/abl_bin/smfmac16_layout_probe > $OUT/smfmac16_layout.txt 2>&1
cat $OUT/smfmac16_layout.txt
