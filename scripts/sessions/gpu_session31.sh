#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "fixed_lambda or region_counts or synthetic" 2>&1 | tail -8
