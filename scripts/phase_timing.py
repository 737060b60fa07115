"""GPU: per-phase cycle breakdown of the transition kernel (pct_debug_phase_timing).
python scripts/phase_timing.py [envs] [steps]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
MODE = sys.argv[3] if len(sys.argv) > 3 else "c2"
if MODE == "c3":
    env = pkg.PctVecEnv(N, continuous=True, sample_left_bound=1.0, sample_right_bound=5.0, seed=4, device="cuda:0", monitor=False)
else:
    env = pkg.PctVecEnv(N, item_set=items, seed=4, device="cuda:0", monitor=False)
env.reset()
rows = torch.empty(N, 9, dtype=torch.float32, device="cuda:0")
for _ in range(200):
    env.policy_hash_rows(rows)
    env.step_rows_device(rows)
torch.cuda.synchronize()
env.phase_timing(True)
per_step_max = []
names = ["load", "drop", "genems", "set", "feas", "obs", "store"]
sub = {8: "set.gen", 9: "set.dedup", 10: "set.match", 11: "set.rebuild"}
prev = np.zeros((N, 16), np.uint64)
for s in range(K):
    env.policy_hash_rows(rows)
    env.step_rows_device(rows)
    if s % 10 == 9:
        cur = env.phase_timing(True)  # read + restart
        tot = cur[:, :7].sum(1).astype(np.float64) / 10
        per_step_max.append((tot.mean(), np.percentile(tot, 99), tot.max()))
t = None
# one more aggregate pass
env.phase_timing(True)
for s in range(K):
    env.policy_hash_rows(rows)
    env.step_rows_device(rows)
acc = env.phase_timing(False).astype(np.float64)
steps = acc[:, 7]
print("envs %d, steps/env %d" % (N, steps.mean()))
tot = acc[:, :7].sum(1) / steps
for i, n in enumerate(names):
    v = acc[:, i] / steps
    print("  %-7s mean %9.0f cyc  p99 %9.0f  max %9.0f   (%.1f%%)" % (n, v.mean(), np.percentile(v, 99), v.max(), 100 * v.mean() / tot.mean()))
for i, n in sub.items():
    v = acc[:, i] / steps
    print("    %-11s mean %9.0f cyc  p99 %9.0f  max %9.0f" % (n, v.mean(), np.percentile(v, 99), v.max()))
print("  total   mean %9.0f cyc  p99 %9.0f  max %9.0f" % (tot.mean(), np.percentile(tot, 99), tot.max()))
pm = np.array(per_step_max)
print("per-10-step windows: mean of env-mean %.0f, mean of env-p99 %.0f, mean of env-max %.0f cycles/step" % tuple(pm.mean(0)))
