"""GPU: per-STEP cycle distribution of the transition kernel (timed build, read back after every launch):
how the slowest env of a launch -- which sets the launch time -- differs from the mean, and how the cycles
scale with the EMS count, the generated tuples and the distinct candidates.
python scripts/step_profile.py [envs] [steps] [c2|c3|c5|c1]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
MODE = sys.argv[3] if len(sys.argv) > 3 else "c2"
items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
if MODE == "c5":
    env = pkg.PctVecEnv(N, continuous=True, container_size=(100, 100, 100), internal_node_holder=200, leaf_node_holder=200,
                        sample_left_bound=5.0, sample_right_bound=25.0, seed=4, device="cuda:0", monitor=False,
                        ems_capacity=384, candidate_capacity=8192)
elif MODE == "c1":
    env = pkg.PctVecEnv(N, setting=1, item_set=items, seed=4, device="cuda:0", monitor=False, lstsq=os.environ.get("PCT_LSTSQ", "gelsd"))
elif MODE == "c3s1":
    env = pkg.PctVecEnv(N, continuous=True, setting=1, container_size=(1, 1, 1), sample_left_bound=0.1, sample_right_bound=0.5, lstsq=os.environ.get("PCT_LSTSQ", "gelsd"),
                        seed=4, device="cuda:0", monitor=False)
elif MODE == "c3":
    env = pkg.PctVecEnv(N, continuous=True, sample_left_bound=1.0, sample_right_bound=5.0, seed=4, device="cuda:0", monitor=False)
else:
    env = pkg.PctVecEnv(N, item_set=items, seed=4, device="cuda:0", monitor=False)
env.reset()
rows = torch.empty(N, 9, dtype=torch.float32, device="cuda:0")
for _ in range(200):
    env.policy_hash_rows(rows); env.step_rows_device(rows)
torch.cuda.synchronize()
env.phase_timing(True)
rec = np.zeros((K, N, 44))
for s in range(K):
    env.policy_hash_rows(rows); env.step_rows_device(rows)
    rec[s] = env.phase_timing(True)
env.phase_timing(False)
names = ["load", "drop", "genems", "set", "feas", "obs", "store"]
tot = rec[:, :, :7].sum(2)  # [K,N]
print("%s: %d envs x %d steps, cycles per step (2.39 GHz shader clock)" % (MODE, N, K))
print("  per-step total: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f" % (
    tot.mean(), *np.percentile(tot, [50, 90, 99, 99.9]), tot.max()))
mx = tot.max(1)
print("  per-launch max over envs: mean %.0f (= %.1f us)  min %.0f  max %.0f ; ratio to mean step %.2f" % (
    mx.mean(), mx.mean() / 2390, mx.min(), mx.max(), mx.mean() / tot.mean()))
am = tot.argmax(1)
worst = rec[np.arange(K), am]  # [K,16]
print("  slowest env of each launch, mean phase cycles:")
for i, n in enumerate(names):
    print("    %-7s %8.0f   (all-env mean %8.0f)" % (n, worst[:, i].mean(), rec[:, :, i].mean()))
for i, n in {8: "set.gen", 9: "set.dedup", 10: "set.match", 11: "set.rebuild"}.items():
    print("      %-11s %8.0f   (all-env mean %8.0f)" % (n, worst[:, i].mean(), rec[:, :, i].mean()))
print("    EMS %.1f (mean %.1f)  distinct %.1f (mean %.1f)  generated %.1f (mean %.1f)" % (
    worst[:, 12].mean(), rec[:, :, 12].mean(), worst[:, 13].mean(), rec[:, :, 13].mean(), worst[:, 14].mean(), rec[:, :, 14].mean()))
extra = {16: "match calls", 17: "match outer rounds", 18: "match longest-walk sum", 19: "cycles inside the matching walk loops",
         20: "contains longest-walk sum", 21: "flushes", 22: "fast-start scalar replay cycles", 23: "rebuilds", 24: "flush hash cycles", 25: "gen: tuple build cycles", 26: "gen: hash cycles", 27: "gen: contains cycles", 28: "gen: pend/ballot cycles", 29: "gen: pair filter cycles"}
for i, n in extra.items():
    print("    %-34s %9.1f   (all-env mean %9.1f)" % (n, worst[:, i].mean(), rec[:, :, i].mean()))
if MODE in ("c1", "c3s1"):  # the stability counters have slots of their own (30..38)
    for i, n in {30: "commit visits", 31: "virtual passes", 32: "virtual tasks", 33: "narrow passes", 34: "lsq k=3", 35: "lsq k=4", 36: "lsq k=5",
                 37: "lsq k>5", 38: "level-0 candidates", 39: "solve rounds (one solve's latency each)", 40: "  ... at level 0 (a round's candidates)", 41: "  ... in the commit walk", 42: "level-0 rounds", 43: "virtual-check calls (batches of 64)"}.items():
        print("    %-34s %9.2f   (all-env mean %9.2f)" % (n, worst[:, i].mean(), rec[:, :, i].mean()))
# least squares: total ~ a + b*E + c*generated + d*distinct
X = np.stack([np.ones(K * N), rec[:, :, 12].ravel(), rec[:, :, 14].ravel(), rec[:, :, 13].ravel()], 1)
coef, *_ = np.linalg.lstsq(X, tot.ravel(), rcond=None)
print("  fit total ~ %.0f + %.1f*EMS + %.2f*generated + %.1f*distinct" % tuple(coef))
d = rec[:, :, 13].ravel()
t = tot.ravel()
for lo, hi in [(0, 5), (5, 19), (19, 77), (77, 150), (150, 307), (307, 2000)]:
    m = (d >= lo) & (d < hi)
    if m.any():
        print("  distinct in [%d,%d): %5.1f%% of steps, mean total %.0f, set %.0f" % (
            lo, hi, 100 * m.mean(), t[m].mean(), rec[:, :, 3].ravel()[m].mean()))

# the single slowest env-steps of the whole run (what a rare long launch is made of)
flat = tot.ravel()
for idx in np.argsort(-flat)[:5]:
    si, ei = divmod(int(idx), N)
    print("  slowest env-step: step %d env %d: %.0f cycles = %.0f us;" % (si, ei, flat[idx], flat[idx] / 2390),
          " ".join("%s %.0f" % (n, rec[si, ei, i]) for i, n in enumerate(names)), "EMS %d boxes? distinct %d" % (rec[si, ei, 12], rec[si, ei, 13]))

# Heavy-first dispatch (pct_order_kernel): how long a launch is when S envs are resident at a time and the rest are handed
# out, in a given order, as slots free up (greedy list scheduling on the measured cycles) -- in workgroup-id order, sorted
# by the previous step's cycles (what the kernel does), by the previous step's EMS count, and by the step's own cycles
# (the best any predictor could do); lower bound max(sum / S, max).
import heapq
SLOTS = int(os.environ.get("PCT_PROFILE_SLOTS", "0")) or {"c1": 1024, "c3s1": 1024, "c3": 2816, "c5": 1280}.get(MODE, N)


def makespan(cost, order, S):
    heap = [0.0] * S
    for e in order:
        heapq.heappush(heap, heapq.heappop(heap) + cost[e])
    return max(heap)


if SLOTS < N:
    res = {"id order": [], "prev cycles": [], "prev EMS": [], "prev cycles x EMS": [], "rank sum of both": [],
           "2-step mean cycles": [], "scaled sum (the kernel's key)": [], "scaled cycles + EMS^2": [], "scaled cycles + 2 EMS": [],
           "own cycles (ideal)": [], "lower bound": []}
    cc = []
    for s in range(1, K):
        cost, prev = tot[s], tot[s - 1]
        cc.append(np.corrcoef(cost, prev)[0, 1])
        res["id order"].append(makespan(cost, np.arange(N), SLOTS))
        res["prev cycles"].append(makespan(cost, np.argsort(-prev, kind="stable"), SLOTS))
        res["prev EMS"].append(makespan(cost, np.argsort(-rec[s - 1, :, 12], kind="stable"), SLOTS))
        res["prev cycles x EMS"].append(makespan(cost, np.argsort(-prev * (1 + rec[s - 1, :, 12]), kind="stable"), SLOTS))
        rk = np.argsort(np.argsort(prev)) + np.argsort(np.argsort(rec[s - 1, :, 12]))
        res["rank sum of both"].append(makespan(cost, np.argsort(-rk, kind="stable"), SLOTS))
        res["2-step mean cycles"].append(makespan(cost, np.argsort(-(prev + tot[max(s - 2, 0)]), kind="stable"), SLOTS))
        pe = rec[s - 1, :, 12]
        kc, ke = prev / (prev.max() + 1), pe / (pe.max() + 1)
        res["scaled sum (the kernel's key)"].append(makespan(cost, np.argsort(-(kc + ke), kind="stable"), SLOTS))
        res["scaled cycles + EMS^2"].append(makespan(cost, np.argsort(-(kc + ke * ke), kind="stable"), SLOTS))
        res["scaled cycles + 2 EMS"].append(makespan(cost, np.argsort(-(kc + 2 * ke), kind="stable"), SLOTS))
        res["own cycles (ideal)"].append(makespan(cost, np.argsort(-cost, kind="stable"), SLOTS))
        res["lower bound"].append(max(cost.sum() / SLOTS, cost.max()))
    print("  list scheduling on %d slots (one wave each; timed build), launch length in k cycles, mean over %d launches:" % (SLOTS, K - 1))
    for k, v in res.items():
        print("    %-30s %8.1f" % (k, np.mean(v) / 1e3))
    print("    correlation of an env's cycles with its previous step's: %.2f" % np.mean(cc))
