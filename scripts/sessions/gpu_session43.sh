#!/bin/bash
mkdir -p gpurun_out
{
python scripts/i8_probe.py 50000 10000 2
python scripts/i8_probe.py 5000 20000 3
python scripts/i8_probe.py 33000 4096 2
python scripts/i8_probe.py 1000 777 2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s43_probe.log
cat gpurun_out/s43_probe.log
