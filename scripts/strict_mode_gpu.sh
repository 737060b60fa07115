#!/bin/bash
# The strict least-squares solver (pct_set_lstsq_mode) on the GPU box, one gpurun call (~6 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 540 -- 'bash scripts/strict_mode_gpu.sh r05'
# 1. the torch-free parity check (scripts/gelsd_gpu_check.py; its wide-flat case alone with -v afterwards: the notice sets)
# 2. tests/test_zz_gpu_gelsd.py
# 3. throughput of the stability workloads in the three modes (bench.py --lstsq), the kernel time by HIP events
# Everything lands under gpurun_out/strict_<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r05}
OUT=gpurun_out/strict_$TAG
mkdir -p $OUT
timeout 200 python scripts/gelsd_gpu_check.py > $OUT/gelsd_gpu_check.txt 2>&1; echo "rc=$?" >> $OUT/gelsd_gpu_check.txt
timeout 200 python scripts/gelsd_gpu_check.py --wide-only -v > $OUT/gelsd_gpu_wide.txt 2>&1; echo "rc=$?" >> $OUT/gelsd_gpu_wide.txt
timeout 400 python -m pytest tests/test_zz_gpu_gelsd.py -x -q -m gpu -s > $OUT/pytest_gelsd.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gelsd.txt
for w in c1 c3s1; do
  for m in jacobi gelsd gelsd_avx2; do
    timeout 200 python bench.py --workload $w --lstsq $m --no-cpu-baseline --steps 600 --warmup 100 > $OUT/bench_${w}_${m}.json 2> $OUT/bench_${w}_${m}.err
  done
done
tail -n 12 $OUT/gelsd_gpu_check.txt $OUT/gelsd_gpu_wide.txt
tail -n 5 $OUT/pytest_gelsd.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M env-steps/s" % (d["value"] / 1e6), "kernel %.1f us" % d["roofline"]["kernel_avg_us"], d["config"]["lstsq"])
    except Exception as e:
        print(f, "ERR", e)
PY
