#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_32
timeout 25 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_32/smoke.txt 2>&1; tail -2 gpurun_out/r4_32/smoke.txt
