#!/bin/bash
# discrete setting-2 parity (KAT, fixtures, baseline scale, overflow retry) + benches
OUT=$PWD/gpurun_out/check2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_scale.py -m gpu -x -q -k "not continuous and not cont" > $OUT/tests.txt 2>&1
echo "rc=$?" >> $OUT/tests.txt
tail -6 $OUT/tests.txt
WORKLOADS="c2 c4 c1 c3s1" bash scripts/r03/cmp_variants.sh "$@"
timeout 200 python bench.py --workload c2 --envs-per-gpu 16384 --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2@16384 %.3f M/s kernel %.1f us' % (d['value']/1e6, d['roofline']['kernel_avg_us']))"
