"""Throughput of the wide-flat stream (flat items on a 20^3 bin: load splits over up to 16 supporters, the retry pass's workspace
class) in the default dgelsd mode -- the case round 4 measured at ~150 env-steps/s with one lane per system."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
from tests.common import make_stream
N, steps = 48, 300
items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
for mode in ("gelsd", "jacobi"):
    env = pkg.PctVecEnv(N, setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50,
                        env_id_base=5, item_stream=make_stream(4242, N, 2048, items), device="cuda:0", lstsq=mode)
    env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.step_hash_policy(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not env.error_flags.any()
    print("wide-flat, lstsq=%s: %d envs x %d steps in %.2f s = %.0f env-steps/s (retry-pass launches with work: %s)" % (
        mode, N, steps, dt, N * steps / dt, env.debug_retry_count(totals=True)[2]))
    env.close()
