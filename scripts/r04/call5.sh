#!/bin/bash
OUT=$PWD/gpurun_out/r04e
mkdir -p $OUT
export PCT_EXPERIMENT=1
b() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/$label.json 2> $OUT/$label.err
  python scripts/r04/bline.py $label $OUT/$label.json
}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b c1 X=1 -- --workload c1 --steps 600 --warmup 100
b c3s1 X=1 -- --workload c3s1 --steps 600 --warmup 100
b c2 X=1 -- --mode epilogue --steps 1000 --warmup 100
timeout 300 python scripts/launch_cliff.py c1 300 --timed > $OUT/cliff_c1.txt 2>&1; head -24 $OUT/cliff_c1.txt
timeout 300 python scripts/launch_cliff.py c3s1 600 > $OUT/cliff_c3s1.txt 2>&1; head -20 $OUT/cliff_c3s1.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_numpy_stream.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1
tail -5 $OUT/pytest_parity.txt
timeout 1200 python -m pytest tests/test_gpu_baseline_scale.py -x -q -m gpu -k "stability or c1 or c3s1 or overflow" > $OUT/pytest_scale_stab.txt 2>&1
tail -5 $OUT/pytest_scale_stab.txt
