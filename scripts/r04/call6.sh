#!/bin/bash
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
export PCT_EXPERIMENT=1
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
b c1 X=1 -- --workload c1 --steps 600 --warmup 100
b c3s1 X=1 -- --workload c3s1 --steps 600 --warmup 100
timeout 300 python scripts/launch_cliff.py c1 300 --timed > $OUT/cliff_c1.txt 2>&1; head -16 $OUT/cliff_c1.txt
timeout 400 python scripts/launch_cliff.py c3s1 600 --timed > $OUT/cliff_c3s1.txt 2>&1; head -40 $OUT/cliff_c3s1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "s1 or s3 or lstsq or flat or notice" > $OUT/pytest_stab.txt 2>&1
tail -3 $OUT/pytest_stab.txt
