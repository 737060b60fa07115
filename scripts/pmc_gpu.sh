#!/bin/bash
# quick PMC passes of the step kernel (steady state: 250 warm-up steps, 100 measured)
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline --steps 100 --warmup 250"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/sq1 -o pmc -- $BENCH > /dev/null 2> $OUT/sq1.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR -d $OUT/sq2 -o pmc -- $BENCH > /dev/null 2> $OUT/sq2.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS -d $OUT/sq3 -o pmc -- $BENCH > /dev/null 2> $OUT/sq3.err
ls $OUT/*
