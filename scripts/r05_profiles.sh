#!/bin/bash
# rocprofv3 trace + PMC of every workload (round 5), the timed-build step profiles
for w in "$@"; do
  bash scripts/profile_gpu.sh r05 $w 2000 100 > /dev/null 2>&1
done
mkdir -p gpurun_out/profiles_r05
for m in c2 c3 c5; do timeout 200 python scripts/step_profile.py $([ $m = c5 ] && echo 2048 || echo 4096) $([ $m = c5 ] && echo 16 || echo 60) $m > gpurun_out/profiles_r05/r05_step_profile_$m.txt 2>&1; done
PCT_LSTSQ=gelsd timeout 200 python scripts/step_profile.py 4096 40 c1 > gpurun_out/profiles_r05/r05_step_profile_c1.txt 2>&1
PCT_LSTSQ=jacobi timeout 200 python scripts/step_profile.py 4096 40 c1 > gpurun_out/profiles_r05/r05_step_profile_c1_jacobi.txt 2>&1
PCT_LSTSQ=gelsd timeout 200 python scripts/step_profile.py 4096 30 c3s1 > gpurun_out/profiles_r05/r05_step_profile_c3s1.txt 2>&1
ls -la gpurun_out/profiles_r05 | head -40
du -sh gpurun_out/prof_r05_* | tail -8
