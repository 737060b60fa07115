#!/bin/bash
for w in "c3 1000 100" "c5 200 60" "c3s1 400 80"; do set -- $w
  bash scripts/profile_gpu.sh r03 $1 $2 $3 2>&1 | tail -1
done
for m in "4096 40 c3" "2048 20 c5"; do set -- $m
  timeout 300 python scripts/step_profile.py $1 $2 $3 > gpurun_out/profiles_r03/r03_step_profile_$3.txt 2>/dev/null
done
