#!/bin/bash
# reference-output tests after the LM contiguity fix + LOCO; bench with the reference's own LMM::Analyze as cpu_baseline
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_reference.py -m gpu -q 2>&1 | tail -15 | tee gpurun_out/ref_tests2.log
( time timeout 420 python bench.py ) > gpurun_out/bench_ref.log 2>&1
grep -v amdgpu.ids gpurun_out/bench_ref.log | tail -8
