#!/bin/bash
OUT=$PWD/gpurun_out/r05_pipes
mkdir -p $OUT
for w in c1 c3s1; do for m in gelsd jacobi; do for p in 1 2 4; do
  timeout 300 python bench.py --workload $w --lstsq $m --pipelines $p --no-cpu-baseline --no-rows-line > $OUT/${w}_${m}_p$p.json 2> $OUT/${w}_${m}_p$p.err
done; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_pipes/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-28s %8.3f M/s  ms/step %.4f  kernel_us %7.1f" % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_us"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
