"""What would more resident waves buy?  Times the transition kernel with a 512-slot candidate
table (5.5 KB LDS per env instead of 13 KB; envs whose set outgrows it are flagged and ignored
here -- timing experiment only)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for cap in (2048, 512):
    env = pkg.PctVecEnv(N, item_set=items, seed=4, device="cuda:0", monitor=False, strict=False, candidate_capacity=cap)
    env.reset()
    rows = torch.empty(N, 9, dtype=torch.float32, device="cuda:0")
    for _ in range(200):
        env.policy_hash_rows(rows); env.step_rows_device(rows)
    torch.cuda.synchronize()
    env.profile_enable(True); env.profile_read()
    t0 = time.perf_counter()
    for _ in range(1000):
        env.policy_hash_rows(rows); env.step_rows_device(rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n, ms = env.profile_read()
    fl = env.error_flags
    print("cand_cap %4d: %.1f us/launch (events), %.2f M steps/s wall, envs flagged %d" % (cap, ms / n * 1e3, N * 1000 / dt / 1e6, int((fl != 0).sum())))
    env.close()
