#!/bin/bash
# lease 5: debug aid for the xlarge/batching bit-equality test, then the whole GPU suite WITHOUT -x (every failure at once)
timeout 300 python scripts/exp/r6_dbg1.py > $OUT/dbg1.txt 2>&1; cat $OUT/dbg1.txt | tail -30
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_all.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/pytest_gpu_all.txt | tail -40
