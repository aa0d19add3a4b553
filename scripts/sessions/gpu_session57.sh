#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_eigh.py -m gpu -q -x 2>&1 | tail -8
