#!/bin/bash
BENCH_EXTRA="--mode slot" PROFILE_SUFFIX=_slot bash scripts/profile_gpu.sh r05 c2 2000 100 > /dev/null 2>&1
BENCH_EXTRA="--lstsq jacobi" PROFILE_SUFFIX=_jacobi bash scripts/profile_gpu.sh r05 c1 2000 100 > /dev/null 2>&1
ls gpurun_out/profiles_r05 | grep -e slot -e jacobi
