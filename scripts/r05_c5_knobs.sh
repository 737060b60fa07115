#!/bin/bash
OUT=$PWD/gpurun_out/r05_c5_knobs
mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-rows-line $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; python -c "
import json; d=json.loads([l for l in open('$OUT/$tag.json') if l.startswith('{')][-1]); print('$tag', '%.3f M' % (d['value']/1e6), 'kernel %.1f us' % d['roofline']['kernel_avg_us'], 'ms/step %.4f' % d['ms_per_step'])" 2>/dev/null || tail -2 $OUT/$tag.err; }
EXTRA="" run base A=1
EXTRA="--ems-capacity 448" run ems448 A=1
EXTRA="--ems-capacity 384" run ems384 A=1
EXTRA="--ems-capacity 576" run ems576 A=1
EXTRA="" run order_cycles PCT_EXPERIMENT=1 PCT_ORDER_MODE=0
EXTRA="" run order_ems PCT_EXPERIMENT=1 PCT_ORDER_MODE=1
EXTRA="" run prio_off PCT_EXPERIMENT=1 PCT_WAVE_PRIO=0
EXTRA="--envs-per-gpu 2304" run n2304 A=1
EXTRA="--envs-per-gpu 1536" run n1536 A=1
