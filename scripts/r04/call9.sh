#!/bin/bash
OUT=$PWD/gpurun_out/r04i
mkdir -p $OUT
export PCT_EXPERIMENT=1
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
b c2_host X=1 -- --mode host --steps 500 --warmup 100
b c2_rows X=1 -- --mode rows --steps 1000 --warmup 100
b c2_fused X=1 -- --mode fused --steps 1000 --warmup 100
b c3_host X=1 -- --workload c3 --mode host --steps 300 --warmup 100
timeout 900 python -m pytest tests/test_rollout.py tests/test_gpu_multiproc.py -x -q -m gpu > $OUT/pytest_a.txt 2>&1; tail -3 $OUT/pytest_a.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vecenv or make_vec or surface or strict or infos or monitor or reset_specific" > $OUT/pytest_b.txt 2>&1; tail -3 $OUT/pytest_b.txt
