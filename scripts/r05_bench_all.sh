#!/bin/bash
# every bench workload once (round 5) -> gpurun_out/r05_bench/<name>.json
OUT=$PWD/gpurun_out/r05_bench
mkdir -p $OUT
B="timeout 400 python bench.py"
$B --workload c2 > $OUT/c2.json 2> $OUT/c2.err
$B --workload c2 --steps 20 --warmup 5 > $OUT/c2_driver_shape.json 2> $OUT/c2_driver_shape.err
for w in c4 c3 c5 c1 c3s1; do $B --workload $w > $OUT/$w.json 2> $OUT/$w.err; done
$B --workload c1 --lstsq jacobi --no-cpu-baseline > $OUT/c1_jacobi.json 2> $OUT/c1_jacobi.err
$B --workload c3s1 --lstsq jacobi --no-cpu-baseline > $OUT/c3s1_jacobi.json 2> $OUT/c3s1_jacobi.err
$B --workload c2 --mode slot --no-cpu-baseline > $OUT/c2_slot.json 2> $OUT/c2_slot.err
$B --workload c2 --mode host --no-cpu-baseline > $OUT/c2_host.json 2> $OUT/c2_host.err
$B --workload c2 --mode host_overlap --no-cpu-baseline > $OUT/c2_host_overlap.json 2> $OUT/c2_host_overlap.err
$B --workload c2 --pipelines 2 --no-cpu-baseline > $OUT/c2_pipelines2.json 2> $OUT/c2_pipelines2.err
$B --workload c2 --pipelines 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c2_pipelines2_driver_shape.json 2> $OUT/c2_pipelines2_driver_shape.err
$B --workload c2 --envs-per-gpu 16384 --no-cpu-baseline > $OUT/c2_16384.json 2> $OUT/c2_16384.err
$B --workload c2 --no-overflow-retry --no-cpu-baseline > $OUT/c2_no_retry.json 2> $OUT/c2_no_retry.err
$B --workload c5 --candidate-capacity 32768 --no-cpu-baseline > $OUT/c5_hbm_table.json 2> $OUT/c5_hbm_table.err
$B --workload c3 --envs-per-gpu 8192 --no-cpu-baseline > $OUT/c3_8192.json 2> $OUT/c3_8192.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_bench/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("rows_mode") or {}
        print("%-34s %8.3f M/s  ms/step %.4f  kernel_us %7.1f  frac %.4f  rows_mode %s  cpu %s" % (
            f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"],
            ("%.2f M" % (r["value"] / 1e6)) if r else "-", (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
