"""GPU: heuristic.py's evaluation (mean utilisation / variance / mean length over finished episodes) with the
heuristic baselines running as in-env policies, plus the step rate.
python scripts/heuristic_eval.py [envs] [episodes] [setting]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
EP = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
SETTING = int(sys.argv[3]) if len(sys.argv) > 3 else 2
items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
for name in pkg.HEURISTICS:
    env = pkg.PctVecEnv(N, setting=SETTING, item_set=items, seed=4, device="cuda:0", monitor=False)
    mean, var, length = pkg.evaluate_heuristic(env, name, EP)
    env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    env.step_heuristic(name, K) if SETTING == 2 else [env.step_heuristic(name, 1) for _ in range(K)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s setting %d: %d episodes on %d envs: utilisation %.4f (var %.5f), %.1f items/episode; %.2f M env-steps/s" % (
        name, SETTING, EP, N, mean, var, length, N * K / dt / 1e6))
    env.close()
