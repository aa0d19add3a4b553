#!/bin/bash
# BASELINE config 3 END TO END from files: n = 20 000, p = 1 000 000, PLINK .bed/.bim/.fam on disk -> first pass ->
# kinship (all SNPs) -> eigendecomposition -> -lmm 1 -> .assoc.txt, one process (tests/cpp/gemma_file_driver.cpp)
mkdir -p gpurun_out /tmp/e2e
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/io_check
g++ -std=c++11 -O2 -Iinclude tests/cpp/gemma_file_driver.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/e2e/driver
( time /tmp/e2e/io_check plinkgen /tmp/e2e/S 20000 1000000 64 ) 2>&1 | grep real
ls -la /tmp/e2e/S.bed
( time timeout 150 /tmp/e2e/driver -bfile /tmp/e2e/S -inproc 1 -lmm 1 -o S -outdir /tmp/e2e ) > gpurun_out/e2e_c3.log 2>&1
cat gpurun_out/e2e_c3.log
( head -3 /tmp/e2e/S.assoc.txt; wc -l /tmp/e2e/S.assoc.txt; nproc ) | tee -a gpurun_out/e2e_c3.log
