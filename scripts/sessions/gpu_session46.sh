#!/bin/bash
mkdir -p gpurun_out
{
python scripts/i8_probe.py
PROBE_GENO=zeros python scripts/i8_probe.py
PROBE_GENO=nomiss python scripts/i8_probe.py
PROBE_U=pow2 python scripts/i8_probe.py
PROBE_U=pow2 PROBE_GENO=zeros python scripts/i8_probe.py
python scripts/i8_probe.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/s46_probe.log
cat gpurun_out/s46_probe.log
