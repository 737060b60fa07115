"""GPU soak: HIP path vs the CPU oracle on millions of env-steps (fused stand-in policy, counter
sampler), comparing observations every `check_every` steps and done/reward every step.
python scripts/soak_parity.py [discrete_s2|discrete_s1|continuous_s2|continuous_s1|continuous_c5|cp|fc] [envs] [steps]
PCT_LSTSQ=gelsd|gelsd_avx2|jacobi selects the stability settings' solver on BOTH sides (default: gelsd, the library's default)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("online-3d-bpp-pct_amd")
from oracle.oracle_lib import OracleVecEnv
from oracle import oracle_lib
LSTSQ = os.environ.get("PCT_LSTSQ", "gelsd")
oracle_lib.set_lstsq_mode({"jacobi": oracle_lib.LSTSQ_JACOBI, "gelsd": oracle_lib.LSTSQ_GELSD, "gelsd_avx2": oracle_lib.LSTSQ_GELSD_AVX2}[LSTSQ])
from tests.common import item_set_range

which = sys.argv[1] if len(sys.argv) > 1 else "discrete_s2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
items = item_set_range(1, 5)
threads = min(os.cpu_count() or 1, 64)
if which == "continuous_c5":
    # BASELINE configs[4] per env: 100^3 bin, 200 / 200 nodes, items U(5, 25) -- the two-wave candidate pipeline's workload
    env = pkg.PctVecEnv(N, setting=2, container_size=(100, 100, 100), continuous=True, sample_left_bound=5.0, sample_right_bound=25.0,
                        internal_node_holder=200, leaf_node_holder=200, seed=17, device="cuda:0", strict=False)
    ora = OracleVecEnv(N, setting=2, container_size=(100, 100, 100), env_kind=1, sample_bounds=(5.0, 25.0), internal_node_holder=200,
                       leaf_node_holder=200, threads=threads)
elif which.startswith("continuous"):
    setting = 1 if which.endswith("s1") else 2
    # setting 1 draws z from {0.1..0.5} (C/bin3D.py:110-112), which is meant for the unit bin (givenData.py:5)
    bin_, lo, hi = ((1, 1, 1), 0.1, 0.5) if setting == 1 else ((10, 10, 10), 1.0, 5.0)
    env = pkg.PctVecEnv(N, setting=setting, container_size=bin_, continuous=True, sample_left_bound=lo,
                        sample_right_bound=hi, seed=17, device="cuda:0", strict=False, lstsq=LSTSQ)
    ora = OracleVecEnv(N, setting=setting, container_size=bin_, env_kind=1, sample_bounds=(lo, hi), threads=threads)
else:
    setting = 1 if which == "discrete_s1" else 2
    lnes = {"cp": ("CP", 3), "fc": ("FC", 4)}.get(which, ("EMS", 0))
    env = pkg.PctVecEnv(N, setting=setting, container_size=(10, 10, 10), item_set=items, seed=17, device="cuda:0",
                        LNES=lnes[0], strict=False, lstsq=LSTSQ)
    ora = OracleVecEnv(N, setting=setting, container_size=(10, 10, 10), item_set=items, lnes=lnes[1], threads=threads)
ora.set_sampler(17)
obs = env.reset(); ora.reset()
t0 = time.time(); bad = 0; eps = 0
for t in range(steps):
    if t % 20 == 0:
        o = obs.cpu().numpy(); r = ora.obs.astype(np.float32)
        if not np.array_equal(o, r):
            e = np.nonzero((o != r).any(1))[0]
            print("OBS MISMATCH step", t, "envs", e[:8], "flags", env.error_flags[e[:8]]); bad = 1; break
    env.step_hash_policy(1); ora.step_hash_policy(1)
    obs, reward, done, infos = env.step_wait()
    eps += int(done.sum())
    if not (np.array_equal(done.astype(np.uint8), ora.done) and np.array_equal(reward[:, 0].numpy(), ora.reward.astype(np.float32))):
        e = np.nonzero(done.astype(np.uint8) != ora.done)[0]
        print("DONE/REWARD MISMATCH step", t, "envs", e[:8], "flags", env.error_flags[e[:8]] if len(e) else None); bad = 1; break
fl = env.error_flags
print("lstsq=" + LSTSQ + " %s: %d envs x %d steps = %d env-steps, %d episodes, mismatches=%d, gpu flags set on %d envs (%s), oracle flags %d, %.1fs" % (
    which, N, t + 1, N * (t + 1), eps, bad, int((fl != 0).sum()), np.unique(fl[fl != 0]).tolist(), int((ora.flags != 0).sum()), time.time() - t0))
