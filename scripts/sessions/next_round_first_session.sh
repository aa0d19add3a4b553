#!/bin/bash
# PLAN for the first gpurun session of round 2 (not run yet: the round-1 GPU minutes were spent).  ~6 GPU-minutes.
#  1. the whole GPU suite -- includes the six file-driven twins added after the last round-1 session
#     (-gxe, -gene, -snps/-notsnp/-km 2, -hwe, multivariate BIMBAM, three traits with missing phenotypes)
#  2. BASELINE config 3 end to end again: the .assoc.txt writer now formats on the thread pool (1.41 s -> expected ~0.2 s)
#  3. the same run through the two-run TEXT hand-off (-gk, then -k): cXX.txt of 20 000 x 20 000 = 4e8 numbers through the
#     threaded WriteMatrix / ReadFile_kin_threaded -- the reference needs minutes for this file
#  4. config 5 (multivariate) again
mkdir -p gpurun_out /tmp/e2e
( time timeout 300 python -m pytest tests -m gpu -q -x --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -14 gpurun_out/pytest_gpu.log
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/io_check
g++ -std=c++11 -O2 -Iinclude tests/cpp/gemma_file_driver.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/driver
/tmp/e2e/io_check plinkgen /tmp/e2e/S 20000 1000000 64
( time timeout 150 /tmp/e2e/driver -bfile /tmp/e2e/S -inproc 1 -lmm 1 -o S -outdir /tmp/e2e ) > gpurun_out/e2e_c3.log 2>&1
cat gpurun_out/e2e_c3.log
( time timeout 200 /tmp/e2e/driver -bfile /tmp/e2e/S -gk 1 -o K -outdir /tmp/e2e ) > gpurun_out/e2e_c3_gk_text.log 2>&1
( time timeout 200 /tmp/e2e/driver -bfile /tmp/e2e/S -k /tmp/e2e/K.cXX.txt -lmm 1 -o T -outdir /tmp/e2e ) >> gpurun_out/e2e_c3_gk_text.log 2>&1
cat gpurun_out/e2e_c3_gk_text.log; ls -la /tmp/e2e/K.cXX.txt
/tmp/e2e/io_check plinkgen /tmp/e2e/M 10000 500000 64 3
( time timeout 120 /tmp/e2e/driver -bfile /tmp/e2e/M -n 1 2 3 -inproc 1 -lmm 1 -o M -outdir /tmp/e2e ) > gpurun_out/e2e_c5.log 2>&1
cat gpurun_out/e2e_c5.log
