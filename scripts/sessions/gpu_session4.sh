#!/bin/bash
mkdir -p gpurun_out
timeout 120 ./scripts/mfma_f64_peak > gpurun_out/mfma_peak.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_parity.py -m gpu -q -rA 2>&1 | grep -E "eigh\[|eigh n=|passed|failed|FAILED|Error|error|assert" | cut -c1-600 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
cat gpurun_out/mfma_peak.log
tail -70 gpurun_out/pytest_gpu.log
