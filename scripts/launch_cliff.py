"""GPU: what makes a stability-setting launch 5-20x longer than the median?

Per launch: its duration (the library's HIP events around the step kernel), and the env whose step took the most
shader-clock cycles -- the work-key word every wave leaves behind the scalars (pct_device.h work_key_end: cycles / 256
<< 12 | live EMS count) is read back after every step.  For the discrete env the timed build adds the phase cycles and
the stability counters (pct_discrete_impl.cuh ST_STAB_*: commit walk visits, virtual-check passes / tasks / narrow
passes, least-squares solves by size) of that env.  The launches beyond 2x the median are listed.

    python scripts/launch_cliff.py <c1|c3s1> [steps] [--timed]
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c3s1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 700
TIMED = "--timed" in sys.argv
W = bench.WORKLOADS[w]
pkg = importlib.import_module("online-3d-bpp-pct_amd")
N = W["envs"]
kw = dict(setting=W["setting"], container_size=W["container"], internal_node_holder=W["I"], leaf_node_holder=W["L"], seed=4,
          device="cuda:0", monitor=False, strict=False)
env = (pkg.PctVecEnv(N, continuous=True, sample_left_bound=W["bounds"][0], sample_right_bound=W["bounds"][1], **kw)
       if W["cont"] else pkg.PctVecEnv(N, item_set=bench.item_set(), **kw))
rows = torch.empty(N, 9, dtype=torch.float32, device="cuda:0")
env.bind_policy_rows(rows)
env.reset()
for _ in range(300):
    env.step_rows_device(rows)
torch.cuda.synchronize()
env.profile_enable(True)
env.profile_read()
if TIMED:
    env.phase_timing(True)
dur = np.zeros(K)
keymax = np.zeros(K, np.int64)
keyarg = np.zeros(K, np.int64)
keymean = np.zeros(K)
emsat = np.zeros(K, np.int64)
boxes = np.zeros(K, np.int64)
retry = np.zeros(K, np.int64)
phase = np.zeros((K, 44))
for s in range(K):
    env.step_rows_device(rows)
    n, ms = env.profile_read()
    dur[s] = ms * 1e3 / max(n, 1)
    keys = env.debug_work_keys()
    cyc = (keys >> 12).astype(np.int64) * 256
    keymax[s], keyarg[s], keymean[s] = cyc.max(), cyc.argmax(), cyc.mean()
    emsat[s] = int(keys[keyarg[s]] & 0xFFF)
    st = env.debug_state(int(keyarg[s]))
    boxes[s] = st["n_boxes"]
    retry[s] = env.debug_retry_count()
    if TIMED:
        rec = env.phase_timing(True)
        phase[s] = rec[keyarg[s]]
env.profile_enable(False)
med = np.median(dur)
print("%s: %d launches of %d envs; step kernel us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f; max/median %.2f" % (
    w, K, N, dur.mean(), med, *np.percentile(dur, [90, 99]), dur.max(), dur.max() / med))
print("  slowest env of a launch: mean %.0f cycles (= %.1f us at 2.4 GHz); mean env %.0f cycles; launches with a queued retry: %d" % (
    keymax.mean(), keymax.mean() / 2400, keymean.mean(), int((retry > 0).sum())))
names = ["load", "drop+commit", "genems", "set", "feas", "obs", "store"]
stat_names = {12: "EMS", 13: "distinct", 30: "commit visits", 31: "virtual passes", 32: "virtual tasks", 33: "narrow passes",
              34: "lsq k=3", 35: "lsq k=4", 36: "lsq k=5", 37: "lsq k>5", 38: "level-0 candidates", 39: "solve rounds"}
order = np.argsort(-dur)
print("  launches beyond 2x the median (%d of %d), longest first:" % (int((dur > 2 * med).sum()), K))
for s in order[:max(12, int((dur > 2 * med).sum()))][:40]:
    line = "    launch %4d: %8.1f us  slowest env %5d: %9d cycles (%.1f us), %3d boxes, %3d EMS, retry queue %d" % (
        s, dur[s], keyarg[s], keymax[s], keymax[s] / 2400, boxes[s], emsat[s], retry[s])
    print(line)
    if TIMED:
        print("        phases: " + "  ".join("%s %d" % (n, phase[s, i]) for i, n in enumerate(names)))
        print("        stats : " + "  ".join("%s %d" % (n, phase[s, i]) for i, n in stat_names.items()))
# consecutive long launches = the lifetime of one episode?
long_ = dur > 2 * med
runs, cur = [], 0
for s in range(K):
    if long_[s]:
        cur += 1
    elif cur:
        runs.append(cur)
        cur = 0
if cur:
    runs.append(cur)
print("  runs of consecutive long launches:", runs)
same = sum(1 for s in range(1, K) if long_[s] and long_[s - 1] and keyarg[s] == keyarg[s - 1])
print("  consecutive long launches with the SAME slowest env: %d" % same)
env.close()
