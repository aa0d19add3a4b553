#!/bin/bash
# closing run of this session: the whole GPU suite (with the file-driven workflows), then the BIMBAM text parser on the
# box's host cores (threads 0 = the reference's way: one thread of strtok + atof)
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests -m gpu -q -x --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -14 gpurun_out/pytest_gpu.log
g++ -std=c++11 -O2 -Iinclude tests/cpp/io_host_check.cpp -Lgemma_amd -lgemma_hip -Wl,-rpath,$PWD/gemma_amd -lz -pthread -o /tmp/io_check
/tmp/io_check genogen /tmp/g.txt 20000 3000
( nproc; ls -la /tmp/g.txt; for t in 0 1 8 32 64 128; do echo "threads $t"; timeout 60 /tmp/io_check genobench /tmp/g.txt 20000 $t; done ) 2>&1 | tee gpurun_out/bimbam_parse.log
