#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench default (driver shape): %.2f M env-steps/s, kernel %.1f us, frac %.4f, rows_mode %.2f M, cpu %.0f, kernel=%s' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['rows_mode']['value']/1e6, d['cpu_baseline']['value'], d['roofline']['kernel'][:60]))"
