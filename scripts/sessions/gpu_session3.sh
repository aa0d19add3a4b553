#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "parity\[|passed|failed|FAILED|Error" | cut -c1-1200 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -60
