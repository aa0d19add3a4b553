#!/bin/bash
# Lease 29: how fast does the GPU box read a .bed-sized file from the page cache, one thread (ifstream, what BlockPrefetch does) against
# 2 / 4 / 8 threads of pread on disjoint parts of each 100 MB block?
cd /tmp && export TMPDIR=/tmp
g++ -O2 -pthread $GRAFT_REPO_ROOT/scripts/exp/r6_29_read_bw.cpp -o /tmp/rd
dd if=/dev/urandom of=/tmp/big bs=1M count=4000 2>/dev/null
nproc
for T in 0 0 1 2 4 8 16; do /tmp/rd /tmp/big 4000 $T; done
rm -f /tmp/big
