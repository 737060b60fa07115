#!/bin/bash
# every bench workload once -> gpurun_out/bench_r02/<workload>.json (+ the phase profile of the timed build)
mkdir -p gpurun_out/bench_r02
for w in c2 c4 c3 c5 c1 c3s1; do
  timeout 400 python bench.py --workload $w > gpurun_out/bench_r02/$w.json 2> gpurun_out/bench_r02/$w.err
done
timeout 200 python bench.py --workload c2 --no-overflow-retry --no-cpu-baseline > gpurun_out/bench_r02/c2_no_retry.json 2>/dev/null
timeout 200 python bench.py --workload c2 --mode fused --no-cpu-baseline > gpurun_out/bench_r02/c2_fused.json 2>/dev/null
timeout 200 python bench.py --workload c2 --envs-per-gpu 16384 --no-cpu-baseline > gpurun_out/bench_r02/c2_16384.json 2>/dev/null
timeout 200 python bench.py --workload c3 --envs-per-gpu 8192 --no-cpu-baseline > gpurun_out/bench_r02/c3_8192.json 2>/dev/null
# the stability workloads with the strict least-squares solver (pct_set_lstsq_mode: dgelsd as the reference's NumPy executes it)
timeout 300 python bench.py --workload c1 --lstsq gelsd --no-cpu-baseline > gpurun_out/bench_r02/c1_gelsd.json 2>/dev/null
timeout 300 python bench.py --workload c3s1 --lstsq gelsd --no-cpu-baseline > gpurun_out/bench_r02/c3s1_gelsd.json 2>/dev/null
timeout 120 python scripts/step_profile.py 4096 100 c2 > gpurun_out/bench_r02/step_profile_c2.txt 2>&1
timeout 120 python scripts/step_profile.py 4096 60 c3 > gpurun_out/bench_r02/step_profile_c3.txt 2>&1
timeout 120 python scripts/step_profile.py 1024 12 c5 > gpurun_out/bench_r02/step_profile_c5.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_r02/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"],
              "frac %.4f" % d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
