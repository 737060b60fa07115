#!/bin/bash
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
export PCT_EXPERIMENT=1
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
S="--workload c5 --steps 100 --warmup 30"
b c5_default X=1 -- $S
b c5_cand8192 X=1 -- $S --candidate-capacity 8192
b c5_cand8192_ems384 X=1 -- $S --candidate-capacity 8192 --ems-capacity 384
b c5_cand8192_ems256 X=1 -- $S --candidate-capacity 8192 --ems-capacity 256
b c5_cand2048_ems256 X=1 -- $S --candidate-capacity 2048 --ems-capacity 256
b c5_ems384 X=1 -- $S --ems-capacity 384
timeout 300 python scripts/step_profile.py 2048 16 c5 > $OUT/step_profile_c5.txt 2>&1; sed -n 2,30p $OUT/step_profile_c5.txt
timeout 300 python scripts/step_profile.py 4096 40 c3 > $OUT/step_profile_c3.txt 2>&1; sed -n 2,30p $OUT/step_profile_c3.txt
