"""GPU: transition-kernel time against the number of envs per launch (latency- or throughput-bound?),
for one or more candidate-table capacities.
python scripts/sweep_envs.py [c2|c3] [cap,cap,...] [n,n,...]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
mode = sys.argv[1] if len(sys.argv) > 1 else "c2"
caps = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
ns = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 1024, 2048, 3072, 4096, 6144, 8192, 16384]
for cap in caps:
    for N in ns:
        if mode == "c3":
            env = pkg.PctVecEnv(N, continuous=True, sample_left_bound=1.0, sample_right_bound=5.0, seed=4, device="cuda:0",
                                monitor=False, strict=False, candidate_capacity=cap)
        else:
            env = pkg.PctVecEnv(N, item_set=items, seed=4, device="cuda:0", monitor=False, strict=False, candidate_capacity=cap)
        env.reset()
        rows = torch.empty(N, 9, dtype=torch.float32, device="cuda:0")
        for _ in range(200):
            env.policy_hash_rows(rows); env.step_rows_device(rows)
        torch.cuda.synchronize()
        env.profile_enable(True); env.profile_read()
        t0 = time.perf_counter()
        K = 400
        for _ in range(K):
            env.policy_hash_rows(rows); env.step_rows_device(rows)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n, ms = env.profile_read()
        fl = env.error_flags
        print("%s cap %5d envs %6d: %8.1f us/launch (events)  %7.2f M env-steps/s wall  flagged %d" % (
            mode, cap, N, ms / n * 1e3, N * K / dt / 1e6, int((fl != 0).sum())), flush=True)
        env.close()
