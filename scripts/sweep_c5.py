"""GPU: C5 (continuous 100^3, 200/200, U(5,25)) step time against the list capacities.
python scripts/sweep_c5.py [envs]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for ems, cand in ((0, 0), (448, 32768), (512, 32768), (640, 32768), (320, 32768)):
    env = pkg.PctVecEnv(N, continuous=True, container_size=(100, 100, 100), internal_node_holder=200, leaf_node_holder=200,
                        sample_left_bound=5.0, sample_right_bound=25.0, seed=4, device="cuda:0", monitor=False, strict=False,
                        ems_capacity=ems, candidate_capacity=cand)
    env.reset()
    env.step_hash_policy(150)
    torch.cuda.synchronize()
    env.profile_enable(True); env.profile_read()
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        env.step_hash_policy(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n, ms = env.profile_read()
    fl = env.error_flags
    print("c5 ems %4d cand %5d envs %5d: %8.1f us/step (events)  %6.3f M env-steps/s wall  flagged %d" % (
        ems, cand, N, ms / K * 1e3, N * K / dt / 1e6, int((fl != 0).sum())), flush=True)
    env.close()
