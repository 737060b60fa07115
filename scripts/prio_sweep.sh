#!/bin/bash
# wave-priority thresholds sweep (PCT_WAVE_PRIO) on C2 / C3: one bench line per setting
mkdir -p gpurun_out
out=gpurun_out/prio_sweep.txt
: > $out
for w in c2 c3; do
  if [ $w = c2 ]; then sets="0 20,26,32 24,30,36 16,22,28 28,32,36 22,30,38"; steps=1500; else sets="0 30,45,60 40,60,80 24,36,48 50,70,90"; steps=600; fi
  for s in $sets; do
    echo "== $w PCT_WAVE_PRIO=$s" >> $out
    PCT_EXPERIMENT=1 PCT_WAVE_PRIO=$s python bench.py --no-cpu-baseline --workload $w --steps $steps --warmup 100 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_avg_us'))
" >> $out
  done
done
python - >> $out <<'PY'
import importlib, numpy as np
pkg = importlib.import_module("online-3d-bpp-pct_amd")
env = pkg.PctVecEnv(1024, setting=2, container_size=(10, 10, 10), continuous=True, sample_from_distribution=True,
                    sample_left_bound=1.0, sample_right_bound=5.0, internal_node_holder=80, leaf_node_holder=50, device="cuda:0")
env.reset()
for t in range(300):
    env.step_hash_policy(1)
env.step_wait()
E = np.array([len(env.debug_state(e)["ems"]) for e in range(0, 1024, 2)])
print("c3 n_ems mean", E.mean(), "pcts 50/75/90/95/99/100", np.percentile(E, [50, 75, 90, 95, 99, 100]))
PY
cat $out
