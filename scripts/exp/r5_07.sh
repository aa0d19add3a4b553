#!/bin/bash
timeout 60 scripts/abl_bin/xcc_mask_probe > $OUT/mask_probe.txt 2>&1; cat $OUT/mask_probe.txt
