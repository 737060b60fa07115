#!/bin/bash
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
export PCT_EXPERIMENT=1
timeout 300 python scripts/launch_cliff.py c1 300 --timed > $OUT/cliff_c1.txt 2>&1; head -70 $OUT/cliff_c1.txt
timeout 300 python scripts/launch_cliff.py c3s1 500 > $OUT/cliff_c3s1.txt 2>&1; head -50 $OUT/cliff_c3s1.txt
timeout 200 python scripts/step_profile.py 4096 60 c2 > $OUT/step_profile_c2.txt 2>&1; sed -n 1,30p $OUT/step_profile_c2.txt
