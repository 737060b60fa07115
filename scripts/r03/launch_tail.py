"""GPU: distribution of the per-step time of a workload (torch events around policy + step), to see what rare long
launches cost.  python scripts/r03/launch_tail.py <c1|c3s1|c3|c5> [steps]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
w = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
W = bench.WORKLOADS[w]
pkg = importlib.import_module("online-3d-bpp-pct_amd")
kw = dict(setting=W["setting"], container_size=W["container"], internal_node_holder=W["I"], leaf_node_holder=W["L"], seed=4,
          device="cuda:0", monitor=False, strict=False)
env = (pkg.PctVecEnv(W["envs"], continuous=True, sample_left_bound=W["bounds"][0], sample_right_bound=W["bounds"][1], **kw)
       if W["cont"] else pkg.PctVecEnv(W["envs"], item_set=bench.item_set(), **kw))
env.reset()
rows = torch.empty(W["envs"], 9, dtype=torch.float32, device="cuda:0")
for _ in range(200):
    env.policy_hash_rows(rows); env.step_rows_device(rows)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
for a, b in ev:
    a.record(); env.policy_hash_rows(rows); env.step_rows_device(rows); b.record()
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print("%s: %d steps, us per step: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (w, K, t.mean(), *np.percentile(t, [50, 90, 99]), t.max()))
srt = np.sort(t)[::-1]
print("  ten longest:", " ".join("%.0f" % x for x in srt[:10]))
print("  mean without the 1%% longest: %.1f us (they cost %.1f %% of the total)" % (srt[K // 100:].mean(), 100 * (1 - srt[K // 100:].sum() / t.sum())))
print("  flags:", int((env.error_flags != 0).sum()))
