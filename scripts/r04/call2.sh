#!/bin/bash
# round 4, GPU call 2: the 8192-env regression (A/B with the r03 library), whole-set hybrid thresholds, non-temporal observation
# stores, where the long stability launches come from
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
V=$PWD/online-3d-bpp-pct_amd/variants
export PCT_EXPERIMENT=1
b() {  # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
S="--steps 1000 --warmup 100"
b main_epilogue X=1 -- --mode epilogue $S
b main_epilogue_noretry X=1 -- --mode epilogue --no-overflow-retry $S
b nt_epilogue PCT_HIP_LIB=$V/libnt.so -- --mode epilogue $S
b nt_epilogue_noretry PCT_HIP_LIB=$V/libnt.so -- --mode epilogue --no-overflow-retry $S
for n in ws100 ws140 ws180; do
  b ${n}_epilogue PCT_HIP_LIB=$V/lib$n.so -- --mode epilogue $S
done
b r03_rows_8192 PCT_HIP_LIB=$V/libr03.so -- --mode rows --envs-per-gpu 8192 $S
b r03_rows_8192_noorder PCT_HIP_LIB=$V/libr03.so PCT_ORDER=0 -- --mode rows --envs-per-gpu 8192 $S
b main_rows_8192 X=1 -- --mode rows --envs-per-gpu 8192 $S
b main_rows_8192_noorder PCT_ORDER=0 -- --mode rows --envs-per-gpu 8192 $S
b main_epilogue_8192 X=1 -- --mode epilogue --envs-per-gpu 8192 $S
b main_epilogue_8192_noorder PCT_ORDER=0 -- --mode epilogue --envs-per-gpu 8192 $S
b main_rows_8192_desync0 X=1 -- --mode rows --envs-per-gpu 8192 --desync 0 --steps 1000 --warmup 200
b main_epilogue_6144 X=1 -- --mode epilogue --envs-per-gpu 6144 $S
b main_epilogue_16384 X=1 -- --mode epilogue --envs-per-gpu 16384 $S
b ws140_epilogue_8192 PCT_HIP_LIB=$V/libws140.so -- --mode epilogue --envs-per-gpu 8192 $S
timeout 300 python scripts/launch_cliff.py c1 300 --timed > $OUT/cliff_c1.txt 2>&1; head -60 $OUT/cliff_c1.txt
timeout 300 python scripts/launch_cliff.py c3s1 500 > $OUT/cliff_c3s1.txt 2>&1; head -40 $OUT/cliff_c3s1.txt
timeout 200 python scripts/step_profile.py 4096 60 c2 > $OUT/step_profile_c2.txt 2>&1
timeout 600 python -m pytest tests/test_rollout.py tests/test_gpu_multiproc.py -x -q -m gpu > $OUT/pytest_small.txt 2>&1; tail -3 $OUT/pytest_small.txt
