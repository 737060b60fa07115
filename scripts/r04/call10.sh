#!/bin/bash
OUT=$PWD/gpurun_out/r04j
mkdir -p $OUT
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
for i in 1 2; do
b c2_host_zc$i X=1 -- --mode host --steps 500 --warmup 100
b c2_host_dev$i X=1 -- --mode host --device-outputs --steps 500 --warmup 100
done
