#!/bin/bash
# kernel variants (scripts/build_variant.py, PCT_FEW_KERNELS): bench the given workloads with each variant library
#   scripts/r05_variants.sh "<workloads>" <variant> [<variant> ...]
OUT=$PWD/gpurun_out/r05_variants
mkdir -p $OUT
export PCT_EXPERIMENT=1
WLS=$1; shift
for v in "$@"; do
  for w in $WLS; do
    PCT_HIP_LIB=$PWD/scripts/r05v/lib$v.so timeout 300 python bench.py --workload $w --no-cpu-baseline --no-rows-line $BENCH_EXTRA > $OUT/${w}_${v}.json 2> $OUT/${w}_${v}.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_variants/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-28s %8.3f M/s  ms/step %.4f  kernel_us %7.1f" % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_us"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
