#!/bin/bash
export PCT_EXPERIMENT=1
mkdir -p gpurun_out/r05_ilv
PCT_HIP_LIB=$PWD/scripts/r05v/libown1.so timeout 900 python -m pytest tests/test_zz_gpu_gelsd.py tests/test_gpu_baseline_scale.py -x -q -m gpu -k "(matches_oracle and not continuous) or c1_setting1 or (every_step and c1) or soak_c1" > gpurun_out/r05_ilv/pytest.txt 2>&1; tail -3 gpurun_out/r05_ilv/pytest.txt
rm -rf gpurun_out/r05_variants
bash scripts/r05_variants.sh "c1" own1
BENCH_EXTRA="--lstsq jacobi" bash scripts/r05_variants.sh "c1" own1 | tail -2
