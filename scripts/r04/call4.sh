#!/bin/bash
OUT=$PWD/gpurun_out/r04d
mkdir -p $OUT
export PCT_EXPERIMENT=1
V=$PWD/online-3d-bpp-pct_amd/variants
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
b c2_rows X=1 -- --mode rows $S
b c2_epilogue_8192 X=1 -- --mode epilogue --envs-per-gpu 8192 $S
b c2_epilogue_16384 X=1 -- --mode epilogue --envs-per-gpu 16384 $S
b c2_p2 X=1 -- --mode epilogue --pipelines 2 $S
b c3 X=1 -- --workload c3 --steps 500 --warmup 100
b c1 X=1 -- --workload c1 --steps 300 --warmup 100
b c3s1 X=1 -- --workload c3s1 --steps 300 --warmup 100
b c5 X=1 -- --workload c5 --steps 100 --warmup 30
for v in $V/libm*.so; do
  [ -e $v ] || continue
  n=$(basename $v .so)
  b ${n}_epilogue PCT_HIP_LIB=$v -- --mode epilogue $S
done
timeout 200 python scripts/step_profile.py 4096 60 c2 > $OUT/step_profile_c2.txt 2>&1; sed -n 2,22p $OUT/step_profile_c2.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1
tail -5 $OUT/pytest_parity.txt
