#!/bin/bash
mkdir -p gpurun_out/r05_bench
timeout 300 python bench.py --workload c5 > gpurun_out/r05_bench/c5.json 2> gpurun_out/r05_bench/c5.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_bench/c5.json') if l.startswith('{')][-1]); print('%.3f M' % (d['value']/1e6), d['roofline']['kernel'][:70], d['roofline']['traffic'])"
