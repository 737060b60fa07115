#!/bin/bash
export PCT_EXPERIMENT=1
export PCT_HIP_LIB=$PWD/scripts/r05v/libtim.so
mkdir -p gpurun_out/r05_tim
PCT_LSTSQ=gelsd timeout 300 python scripts/step_profile.py 4096 40 c1 > gpurun_out/r05_tim/step_profile_c1.txt 2>&1
grep -E "per-launch max|solve rounds|level 0|commit walk|level-0 rounds|virtual-check calls|virtual passes|lsq k" gpurun_out/r05_tim/step_profile_c1.txt
