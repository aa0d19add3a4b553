#!/bin/bash
# file-driven workflows again with the prefetching feeders, then the END-TO-END run SURVEY 8d asks for: PLINK files on
# disk -> first pass -> kinship -> eigendecomposition -> -lmm 1 -> .assoc.txt in one process at n = 20 000, 100 000 SNPs
mkdir -p gpurun_out /tmp/e2e
timeout 120 python -m pytest tests/test_gpu_workflow_files.py tests/test_gpu_host_mirror.py -q -x 2>&1 | tail -3 | tee gpurun_out/wf2.log
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/io_check
g++ -std=c++11 -O2 -Iinclude tests/cpp/gemma_file_driver.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/driver
( time /tmp/e2e/io_check plinkgen /tmp/e2e/S 20000 100000 64 ) 2>&1 | grep real
ls -la /tmp/e2e/S.bed
( time timeout 170 /tmp/e2e/driver -bfile /tmp/e2e/S -inproc 1 -lmm 1 -o S -outdir /tmp/e2e ) > gpurun_out/e2e_n20000.log 2>&1
cat gpurun_out/e2e_n20000.log
head -3 /tmp/e2e/S.assoc.txt | tee -a gpurun_out/e2e_n20000.log; wc -l /tmp/e2e/S.assoc.txt | tee -a gpurun_out/e2e_n20000.log
