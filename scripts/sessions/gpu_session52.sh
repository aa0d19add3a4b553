#!/bin/bash
python scripts/host_path_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
