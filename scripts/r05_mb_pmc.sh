#!/bin/bash
OUT=$PWD/gpurun_out/r05_mb_pmc
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for cfg in "3,1,4,1" "3,2,1,1" "3,0,1,1" "4,2,1,1" "4,1,4,1"; do
  tag=$(echo $cfg | tr ',' '_')
  MB_ONLY=$cfg timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$tag -o p -- python $REPO/scripts/mb_gelsd.py --quick > $OUT/$tag.txt 2>&1
  python - $OUT/$tag $cfg <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if "mb_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    w = acc.get("SQ_WAVES", 1) or 1
    print(sys.argv[2], {k: round(v / w) for k, v in acc.items()}, "waves", w)
PY
done
find $OUT -size +2M -delete
