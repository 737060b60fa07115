#!/bin/bash
# final round-3 profiles of every bench workload (kernel trace + PMC passes), summaries into gpurun_out/profiles_r03/
for w in "c2 2000 100" "c3 1000 100" "c5 200 60" "c1 600 100" "c3s1 400 80" "c4 1000 100"; do set -- $w
  bash scripts/profile_gpu.sh r03 $1 $2 $3 2>&1 | tail -1
done
for m in "4096 60 c2" "4096 40 c1" "4096 40 c3" "2048 20 c5"; do set -- $m
  timeout 300 python scripts/step_profile.py $1 $2 $3 > gpurun_out/profiles_r03/r03_step_profile_$3.txt 2>/dev/null
done
ls gpurun_out/profiles_r03/
