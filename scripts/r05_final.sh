#!/bin/bash
# final state: GPU suite, then the bench lines and the C3 / C5 / c1 profiles that changed since the first r05 profile pass
PYTEST_EXTRA="" bash scripts/r05_gputests.sh | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/r05_bench_all.sh | tail -20
bash scripts/r05_profiles.sh c3 c5 c1 c3s1 > /dev/null 2>&1
timeout 600 python scripts/mb_gelsd.py > gpurun_out/profiles_r05/r05_microbench_gelsd.txt 2>&1
tail -3 gpurun_out/profiles_r05/r05_microbench_gelsd.txt
