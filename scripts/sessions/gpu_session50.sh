#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -k "gxe or gene" 2>&1 | tail -12
