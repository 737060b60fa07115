#!/bin/bash
# One parameterised gpurun call script (replaces the per-experiment one-offs of earlier rounds):
#   scripts/gpu_round.sh <tag> <what>...      what: tests | smoke | bench [workloads..] | modes | mb | prof <workload> | soak
# Everything lands under gpurun_out/<tag>/ ; summaries worth judging are copied into profiles/ by hand afterwards.
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
B="timeout 400 python bench.py"
summ() {
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("rows_mode") or {}
        print("%-36s %8.3f M/s  ms/step %.4f  kernel_us %7.1f  frac %.4f  rows %s  cpu %s" % (
            f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"],
            ("%.2f M" % (r["value"] / 1e6)) if r else "-", (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
}
while [ $# -gt 0 ]; do
  case $1 in
    tests) timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -x -q ${PYTEST_EXTRA} > $OUT/gputest.txt 2>&1; tail -5 $OUT/gputest.txt;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt;;
    bench)
      shift; WL=""
      while [ $# -gt 0 ] && [[ $1 =~ ^c[0-9] ]]; do WL="$WL $1"; shift; done
      [ -z "$WL" ] && WL="c2 c4 c3 c5 c1 c3s1"
      for w in $WL; do $B --workload $w $BENCH_EXTRA > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
      $B --workload c2 --steps 20 --warmup 5 $BENCH_EXTRA > $OUT/bench_c2_driver_shape.json 2> $OUT/bench_c2_driver_shape.err
      summ; continue;;
    modes)
      for m in slot host host_overlap; do $B --workload c2 --mode $m --no-cpu-baseline > $OUT/bench_c2_$m.json 2> $OUT/bench_c2_$m.err; done
      $B --workload c2 --no-overflow-retry --no-cpu-baseline > $OUT/bench_c2_no_retry.json 2> $OUT/bench_c2_no_retry.err
      $B --workload c2 --envs-per-gpu 16384 --no-cpu-baseline > $OUT/bench_c2_16384.json 2> $OUT/bench_c2_16384.err
      for w in c1 c3s1; do for n in 8192 16384; do $B --workload $w --envs-per-gpu $n --no-cpu-baseline > $OUT/bench_${w}_$n.json 2> $OUT/bench_${w}_$n.err; done; done
      for w in c1 c3s1; do $B --workload $w --lstsq jacobi --no-cpu-baseline > $OUT/bench_${w}_jacobi.json 2> $OUT/bench_${w}_jacobi.err; done
      summ;;
    mb) MB_PROF=1 timeout 900 python scripts/mb_gelsd.py $MB_ARGS > $OUT/mb_gelsd.txt 2>&1; tail -40 $OUT/mb_gelsd.txt;;
    prof) shift; bash scripts/profile_gpu.sh $TAG $1 > $OUT/prof_$1.log 2>&1; tail -3 $OUT/prof_$1.log;;
    stepprof) shift; timeout 300 python scripts/step_profile.py ${STEPPROF_ARGS:-4096 100} $1 > $OUT/step_profile_$1.txt 2>&1; tail -30 $OUT/step_profile_$1.txt;;
    soak) timeout 1500 python scripts/soak_parity.py $SOAK_ARGS > $OUT/soak.txt 2>&1; tail -8 $OUT/soak.txt;;
    *) echo "unknown: $1";;
  esac
  shift
done
