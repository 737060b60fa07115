#!/bin/bash
export PCT_EXPERIMENT=1
export PCT_HIP_LIB=$PWD/scripts/r05v/libown.so
mkdir -p gpurun_out/r05_own
timeout 900 python -m pytest tests/test_zz_gpu_gelsd.py -x -q -m gpu -k "matches_oracle" > gpurun_out/r05_own/pytest.txt 2>&1; tail -4 gpurun_out/r05_own/pytest.txt
for w in c1 c3s1; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-rows-line > gpurun_out/r05_own/$w.json 2> gpurun_out/r05_own/$w.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_own/$w.json') if l.startswith('{')][-1]); print('$w', d['value']/1e6, d['roofline']['kernel_avg_us'])"; done
