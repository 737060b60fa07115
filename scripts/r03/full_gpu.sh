#!/bin/bash
# the driver's round-end checks, run by the builder: the whole GPU suite, smoke, every bench workload
OUT=$PWD/gpurun_out/full
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1
echo "rc=$?" >> $OUT/gputest.txt
tail -5 $OUT/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in c2 c4 c3 c5 c1 c3s1; do
  timeout 400 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
timeout 200 python bench.py --workload c2 --envs-per-gpu 16384 --no-cpu-baseline > $OUT/bench_c2_16384.json 2>/dev/null
timeout 200 python bench.py --workload c2 --pipelines 2 --no-cpu-baseline > $OUT/bench_c2_pipelines2.json 2>/dev/null
timeout 200 python bench.py --workload c2 --mode host --no-cpu-baseline > $OUT/bench_c2_host.json 2>/dev/null
timeout 200 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_driver_shape.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/full/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"],
              "frac %.4f" % d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
