#!/bin/bash
# bench c1 / c3s1 with the main library and with each variant under online-3d-bpp-pct_amd/build/v/lib<name>.so
OUT=$PWD/gpurun_out/cmp
mkdir -p $OUT
for v in main "$@"; do
  LIBV=""
  [ "$v" != main ] && LIBV=$PWD/online-3d-bpp-pct_amd/build/v/lib$v.so
  for w in ${WORKLOADS:-c1 c3s1}; do
    PCT_HIP_LIB=$LIBV timeout 200 python bench.py --workload $w --steps ${STEPS:-300} --warmup 100 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
    print("$v $w", "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
except Exception as e:
    print("$v $w ERR", e, open("$OUT/b.err").read()[-300:])
PY
  done
done
