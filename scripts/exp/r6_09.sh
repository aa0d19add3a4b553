#!/bin/bash
timeout 600 python scripts/exp/r6_dbg2.py > $OUT/dbg2.txt 2>&1; grep -v "Warning\|warn" $OUT/dbg2.txt | tail -40
timeout 900 python -m pytest tests -m gpu -q -k "workspace_pool or sharded_backtransformation" > $OUT/pytest_fix.txt 2>&1; tail -6 $OUT/pytest_fix.txt
