#!/bin/bash
timeout 120 scripts/abl_bin/alloc_probe > $OUT/alloc_probe.txt 2>&1; cat $OUT/alloc_probe.txt
