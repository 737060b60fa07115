#!/bin/bash
# the stability workloads with the lane-group dgelsd (round 5) + the strict-mode GPU tests
OUT=$PWD/gpurun_out/r05_gelsd_step2
mkdir -p $OUT
for w in c1 c3s1; do
  for m in gelsd jacobi; do
    timeout 400 python bench.py --workload $w --lstsq $m --no-cpu-baseline > $OUT/bench_${w}_${m}.json 2> $OUT/bench_${w}_${m}.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_gelsd_step2/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_zz_gpu_gelsd.py -x -q -m gpu -k "not matches_oracle" > $OUT/pytest_gelsd.txt 2>&1
tail -5 $OUT/pytest_gelsd.txt
