#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of one bench workload.
# Outputs under gpurun_out/prof_<tag>_<workload>/ ; summaries are copied into profiles/ by
# scripts/collect_profiles.py afterwards.   scripts/profile_gpu.sh <tag> <workload> [trace-steps]
TAG=${1:-r02}
WL=${2:-c2}
TSTEPS=${3:-2000}
PSTEPS=${4:-100}
REPO=$PWD
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}${PROFILE_SUFFIX:-}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline --no-rows-line --repeats 1 --workload $WL $BENCH_EXTRA"  # (one timed region: collect_profiles.py counts dispatches per step)  # (BENCH_EXTRA: e.g. --lstsq jacobi, --mode slot)
SUF=${PROFILE_SUFFIX:-}  # names the variant in the file names (e.g. _jacobi)
cd /tmp
# 1. kernel trace + stats over the same command as the bench line
timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $OUT/trace -o trace -- $BENCH --steps $TSTEPS --warmup 200 > $OUT/trace_bench.json 2> $OUT/trace.err
# 2. PMC passes (own runs, kernel-trace only)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o pmc -- $BENCH --desync 0 --steps $PSTEPS --warmup 100 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR -d $OUT/pmc_sq2 -o pmc -- $BENCH --desync 0 --steps $PSTEPS --warmup 100 > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH --desync 0 --steps $PSTEPS --warmup 100 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH --desync 0 --steps $PSTEPS --warmup 100 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# summaries are made here, on the GPU box (the databases are too large to travel back whole)
cd $REPO
ENVS=$(python -c "import bench; print(bench.WORKLOADS['$WL']['envs'])")
ALG=$(python -c "import bench; print(bench.alg_bytes(bench.WORKLOADS['$WL']))")
PCT_PROFILE_DST=$REPO/gpurun_out/profiles_$TAG PROFILE_SUFFIX=$SUF python scripts/collect_profiles.py $TAG $WL $ENVS $ALG $TSTEPS $PSTEPS > $OUT/collect.log 2>&1
# the raw rocprofv3 kernel-stats CSV of the traced run travels with the summaries (keep the raw profiler output)
cp $OUT/trace/*kernel_stats.csv $REPO/gpurun_out/profiles_$TAG/${TAG}_trace_${WL}${SUF}_kernel_stats.csv 2>/dev/null
cd $OUT
# keep the merge-back bounded: the per-dispatch PMC tables and the trace CSV of a 2000-step run are a few MB each
find . -size +1M -delete  # (the rocpd databases and the per-dispatch trace / counter tables: 20-30 MB per workload; the stats CSVs stay)
du -sh .
