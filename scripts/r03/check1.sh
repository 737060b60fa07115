#!/bin/bash
# first GPU check of the restructured stability kernels: the stability / heuristic / strict-mode tests, then short benches
OUT=$PWD/gpurun_out/check1
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "s1 or s3 or stab or heur or flat or stream or rollout or baseline" > $OUT/tests_stab.txt 2>&1
echo "rc=$?" >> $OUT/tests_stab.txt
tail -15 $OUT/tests_stab.txt
for w in c1 c3s1 c2 c3; do
  timeout 300 python bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_$w.json") if l.startswith("{")][-1])
    print("$w", "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
except Exception as e:
    print("$w ERR", e, open("$OUT/bench_$w.err").read()[-600:])
PY
done
