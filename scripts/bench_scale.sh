#!/bin/bash
# The 1 -> 8-GPU curve in one command: bench.py at N = 1, 2, 4, 8 for the headline workload (c2:
# configs[1], 4096 envs per GPU) and the configs[3] slice (c4: 8192 envs per GPU -- `--gpus 8 --workload c4` IS the 65 536-env
# line).  One JSON line per (workload, N) on stdout, a copy under gpurun_out/scale/.  Envs shard by global id, no collective
# on the step path ("scaling": "weak").
#     scripts/bench_scale.sh [steps] [warmup]        (N is capped at the GPUs this node shows)
STEPS=${1:-2000}
WARMUP=${2:-200}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
OUT=gpurun_out/scale
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29541
for W in c2 c4; do
  for N in 1 2 4 8; do
    [ "$N" -gt "$NGPU" ] && continue
    if [ "$N" -eq 1 ]; then
      python bench.py --gpus 1 --workload $W --steps $STEPS --warmup $WARMUP > $OUT/${W}_n$N.json 2> $OUT/${W}_n$N.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --workload $W --steps $STEPS --warmup $WARMUP > $OUT/${W}_n$N.json 2> $OUT/${W}_n$N.err
      PORT=$((PORT + 1))
    fi
    grep '^{' $OUT/${W}_n$N.json | tail -1
  done
done
