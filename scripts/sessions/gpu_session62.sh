#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -k "int8 or state_errors or plink" 2>&1 | tail -4
