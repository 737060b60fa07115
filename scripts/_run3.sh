bash scripts/gpu_round.sh r06d tests
bash scripts/gpu_round.sh r06d bench c1 c3s1
STEPPROF_ARGS="4096 60" bash scripts/gpu_round.sh r06d stepprof c1 > /dev/null
STEPPROF_ARGS="4096 40" bash scripts/gpu_round.sh r06d stepprof c3s1 > /dev/null
