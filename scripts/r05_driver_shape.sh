#!/bin/bash
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('driver shape: %.2f M  rows %.2f M  kernel %.1f us  timed launches %d' % (d['value']/1e6, d['rows_mode']['value']/1e6, d['roofline']['kernel_avg_us'], d['roofline']['launches_timed']))"; done
