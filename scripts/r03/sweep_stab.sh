#!/bin/bash
# stability kernels: walk-queue / hull-workspace sweep (capacities are read from the environment at pct_create)
OUT=$PWD/gpurun_out/sweep_stab
mkdir -p $OUT
for cfg in ${CFGS:-"112 8960" "160 8960" "256 8960"}; do
  set -- $cfg
  for w in ${WORKLOADS:-c1 c3s1}; do
    PCT_STAB_Q=$1 PCT_STAB_WS=$2 PCT_STAB_SP=$3 PCT_STAB_PP=$4 timeout 200 python bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
    print("Q=$1 WS=$2 SP=$3 PP=$4 $w", "%.3f M/s" % (d["value"] / 1e6), "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
except Exception as e:
    print("Q=$1 WS=$2 $w ERR", e, open("$OUT/b.err").read()[-300:])
PY
  done
done
