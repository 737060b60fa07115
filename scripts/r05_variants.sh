#!/bin/bash
# kernel variants of the discrete stability translation unit (scripts/build_variant.py, PCT_FEW_KERNELS), bench c1 in both solver modes
OUT=$PWD/gpurun_out/r05_variants
mkdir -p $OUT
export PCT_EXPERIMENT=1
for v in "$@"; do
  for m in gelsd jacobi; do
    PCT_HIP_LIB=$PWD/scripts/r05v/lib$v.so timeout 300 python bench.py --workload c1 --lstsq $m --no-cpu-baseline > $OUT/c1_${v}_${m}.json 2> $OUT/c1_${v}_${m}.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_variants/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
