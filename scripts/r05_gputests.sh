#!/bin/bash
OUT=$PWD/gpurun_out/r05_gputests
mkdir -p $OUT
timeout 3000 python -m pytest tests/ -q -m gpu --durations=10 $PYTEST_EXTRA > $OUT/pytest_gpu.txt 2>&1
tail -40 $OUT/pytest_gpu.txt
