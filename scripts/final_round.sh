#!/bin/bash
# The round's closing measurements on the GPU box, in three gpurun calls (each bounded):
#   scripts/final_round.sh <tag> tests | bench | profiles | soak
TAG=${1:-r06}; WHAT=${2:-tests}
O=$PWD/gpurun_out/${TAG}_final
mkdir -p $O
case $WHAT in
  tests)
    timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gputest.txt 2>&1; tail -12 $O/gputest.txt
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt;;
  bench)
    bash scripts/gpu_round.sh ${TAG}_final bench
    bash scripts/gpu_round.sh ${TAG}_final modes
    MB_PROF=1 timeout 900 python scripts/mb_gelsd.py > $O/mb_gelsd.txt 2>&1; grep -c "bit-exact" $O/mb_gelsd.txt; tail -1 $O/mb_gelsd.txt;;
  profiles)
    for w in c2 c3 c5 c1 c3s1 c4; do bash scripts/profile_gpu.sh $TAG $w > $O/prof_$w.log 2>&1; tail -1 $O/prof_$w.log; done
    BENCH_EXTRA="--mode slot" PROFILE_SUFFIX=_slot bash scripts/profile_gpu.sh $TAG c2 > $O/prof_c2_slot.log 2>&1
    for w in c2 c3 c5 c1 c3s1; do
      case $w in c5) A="2048 16";; c3s1) A="4096 40";; *) A="4096 60";; esac
      timeout 300 python scripts/step_profile.py $A $w > $O/step_profile_$w.txt 2>&1
    done
    ls gpurun_out/profiles_$TAG | head -40;;
  soak)
    for spec in "discrete_s2 4096 2000" "discrete_s1 4096 2000" "continuous_s2 4096 1000" "continuous_s1 2048 1500" "continuous_c5 2048 200"; do
      timeout 900 python scripts/soak_parity.py $spec >> $O/soak.txt 2>&1; tail -1 $O/soak.txt
    done;;
esac
