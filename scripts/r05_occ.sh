#!/bin/bash
# occupancy experiment: the 2-waves-per-SIMD variant with the per-env LDS shrunk through the experiment knobs so that 6 / 8 envs fit a CU
OUT=$PWD/gpurun_out/r05_occ
mkdir -p $OUT
export PCT_EXPERIMENT=1
export TMPDIR=/tmp
REPO=$PWD
run() {  # tag lib mode extra-env...
  local tag=$1 lib=$2 mode=$3; shift 3
  env "$@" PCT_HIP_LIB=$REPO/scripts/r05v/lib$lib.so timeout 300 python bench.py --workload c1 --lstsq $mode --no-cpu-baseline --steps 1000 > $OUT/${tag}_${mode}.json 2> $OUT/${tag}_${mode}.err
}
lds() {  # tag lib extra-env...
  local tag=$1 lib=$2; shift 2
  (cd /tmp && env "$@" PCT_HIP_LIB=$REPO/scripts/r05v/lib$lib.so timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o t -- python $REPO/bench.py --workload c1 --no-cpu-baseline --steps 8 --warmup 2 --desync 10 > /dev/null 2> $OUT/tr_$tag.err)
  python - $OUT/tr_$tag $tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "pct_discrete_kernel" in r["Kernel_Name"] and "Li0ELb0ELb1" in r["Kernel_Name"].replace(" ", "") or "5, 0, false, true" in r["Kernel_Name"]]
    if rows:
        r = rows[len(rows) // 2]
        print(sys.argv[2], "LDS", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"), "vgpr", r.get("VGPR_Count"), "accum", r.get("Accum_VGPR_Count"), "grid", r.get("Grid_Size"))
PY
}
K6="PCT_CAND_CAP=128 PCT_STAB_SP=48 PCT_STAB_PP=128 PCT_STAB_WS=4480 PCT_STAB_Q=96"
K8="PCT_CAND_CAP=128 PCT_STAB_SP=40 PCT_STAB_PP=112 PCT_STAB_WS=2240 PCT_STAB_Q=64"
lds base fbase
lds k6 coop2 $K6
lds k8 coop2 $K8
for m in gelsd jacobi; do
  run base_fbase fbase $m
  run k6_coop1 coop1 $m $K6
  run k6_coop2 coop2 $m $K6
  run k8_coop1 coop1 $m $K8
  run k8_coop2 coop2 $m $K8
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_occ/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "%.3f M/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_us %.1f" % d["roofline"]["kernel_avg_us"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
find $OUT -size +2M -delete
