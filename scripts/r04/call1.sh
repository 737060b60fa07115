#!/bin/bash
# round 4, GPU call 1: any-order probe, parity subset with the kernarg build, C2 bench matrix (dispatch structure), phase profile
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
V=$PWD/online-3d-bpp-pct_amd/variants
scripts/bin/anyorder_probe > $OUT/anyorder.txt 2>&1
cat $OUT/anyorder.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
b() {  # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
S="--steps 1000 --warmup 100"
b r03_rows PCT_HIP_LIB=$V/libr03.so -- --mode rows $S
b new_rows X=1 -- --mode rows $S
b new_rows_nooverlap PCT_RETRY_OVERLAP=0 -- --mode rows $S
b new_epilogue X=1 -- --mode epilogue $S
b new_epilogue_nooverlap PCT_RETRY_OVERLAP=0 -- --mode epilogue $S
b new_epilogue_noretry X=1 -- --mode epilogue --no-overflow-retry $S
b new_epilogue_rb4 PCT_RETRY_BLOCKS=4 -- --mode epilogue $S
b new_fused X=1 -- --mode fused $S
b new_epilogue_driver X=1 -- --mode epilogue --steps 20 --warmup 5
b r03_rows_driver PCT_HIP_LIB=$V/libr03.so -- --mode rows --steps 20 --warmup 5 --desync 0
b new_epilogue_8192 X=1 -- --mode epilogue --envs-per-gpu 8192 $S
b new_epilogue_16384 X=1 -- --mode epilogue --envs-per-gpu 16384 $S
b new_c3 X=1 -- --workload c3 --steps 500 --warmup 100
b new_c1 X=1 -- --workload c1 --steps 300 --warmup 100
b new_c3s1 X=1 -- --workload c3s1 --steps 300 --warmup 100
b new_c5 X=1 -- --workload c5 --steps 100 --warmup 30
for v in $V/libws*.so; do
  [ -e $v ] || continue
  n=$(basename $v .so)
  b ${n}_epilogue PCT_HIP_LIB=$v -- --mode epilogue $S
done
timeout 200 python scripts/step_profile.py 4096 60 c2 > $OUT/step_profile_c2.txt 2>&1
head -24 $OUT/step_profile_c2.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_rollout.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1
tail -5 $OUT/pytest_parity.txt
