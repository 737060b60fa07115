#!/bin/bash
# round 5, first GPU minutes: the stability workloads in both solver modes, bench line + kernel trace (raw CSVs kept)
OUT=$PWD/gpurun_out/r05_gelsd_base
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for w in c1 c3s1; do
  for m in jacobi gelsd; do
    timeout 400 python bench.py --workload $w --lstsq $m --no-cpu-baseline > $OUT/bench_${w}_${m}.json 2> $OUT/bench_${w}_${m}.err
  done
done
cd /tmp
for w in c1 c3s1; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_${w}_gelsd -o trace --output-format csv -- python $REPO/bench.py --workload $w --lstsq gelsd --no-cpu-baseline --steps 500 > $OUT/trace_${w}_gelsd.json 2> $OUT/trace_${w}_gelsd.err
done
cd $OUT
find . -size +8M -delete
du -sh .
tail -n 2 bench_*.json | cut -c1-400
