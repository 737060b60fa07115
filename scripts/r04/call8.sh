#!/bin/bash
OUT=$PWD/gpurun_out/r04h
mkdir -p $OUT
export PCT_EXPERIMENT=1
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
S="--steps 1000 --warmup 100"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b c2_epilogue X=1 -- --mode epilogue $S
b c2_epilogue_noretry X=1 -- --mode epilogue --no-overflow-retry $S
b c2_driver X=1 -- --steps 20 --warmup 5
b c2_8192 X=1 -- --mode epilogue --envs-per-gpu 8192 $S
b c2_16384 X=1 -- --mode epilogue --envs-per-gpu 16384 $S
b c1 X=1 -- --workload c1 --steps 300 --warmup 100
timeout 200 python scripts/step_profile.py 4096 60 c2 > $OUT/step_profile_c2.txt 2>&1; sed -n 2,26p $OUT/step_profile_c2.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not s1 and not s3" > $OUT/pytest_parity.txt 2>&1
tail -3 $OUT/pytest_parity.txt
