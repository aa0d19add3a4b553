#!/bin/bash
# file-driven workflows incl. -loco and the multivariate LMM; then BASELINE config 5 END TO END from files:
# mvLMM, n = 10 000, p = 500 000, 3 phenotypes, PLINK on disk -> .assoc.txt in one process
mkdir -p gpurun_out /tmp/e2e
timeout 150 python -m pytest tests/test_gpu_workflow_files.py -q -x 2>&1 | tail -4 | tee gpurun_out/wf4.log
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/io_check
g++ -std=c++11 -O2 -Iinclude tests/cpp/gemma_file_driver.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/driver
( time /tmp/e2e/io_check plinkgen /tmp/e2e/M 10000 500000 64 3 ) 2>&1 | grep real
ls -la /tmp/e2e/M.bed
( time timeout 120 /tmp/e2e/driver -bfile /tmp/e2e/M -n 1 2 3 -inproc 1 -lmm 1 -o M -outdir /tmp/e2e ) > gpurun_out/e2e_c5.log 2>&1
cat gpurun_out/e2e_c5.log
( head -3 /tmp/e2e/M.assoc.txt | cut -c1-260; wc -l /tmp/e2e/M.assoc.txt ) | tee -a gpurun_out/e2e_c5.log
