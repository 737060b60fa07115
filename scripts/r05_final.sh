#!/bin/bash
# final state: GPU suite, smoke, the bench lines, the profiles of the stability workloads, a soak of both envs
PYTEST_EXTRA="" bash scripts/r05_gputests.sh | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/r05_bench_all.sh | tail -20
for w in c1 c3s1; do for m in gelsd jacobi; do timeout 300 python bench.py --workload $w --lstsq $m --pipelines 2 --no-cpu-baseline --no-rows-line > gpurun_out/r05_bench/${w}_${m}_pipelines2.json 2>/dev/null; done; done
bash scripts/r05_profiles.sh c1 c3s1 > /dev/null 2>&1
BENCH_EXTRA="--lstsq jacobi" PROFILE_SUFFIX=_jacobi bash scripts/profile_gpu.sh r05 c1 2000 100 > /dev/null 2>&1
{ PCT_LSTSQ=gelsd timeout 600 python scripts/soak_parity.py discrete_s1 4096 3000; PCT_LSTSQ=gelsd timeout 600 python scripts/soak_parity.py continuous_s1 4096 2000; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/profiles_r05/r05_soak_parity_final.txt
