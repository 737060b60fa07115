#!/bin/bash
timeout 600 python scripts/wide_flat_rate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/profiles_r05/r05_wide_flat_rate.txt
