#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/exp/r6_31.py 2>&1 | grep -v "^$" | tail -12
