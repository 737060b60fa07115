"""Times the REFERENCE's own CPU path -- envs.make_vec_envs -> ShmemVecEnv(context='fork') wrapped in
VecPyTorch(cpu) (envs.py:75-116,159-182; wrapper/shmem_vec_env.py:20-156), one forked worker per env --
and, for BASELINE configs[0], DummyVecEnv with one env (wrapper/dummy_vec_env.py:45-62), in THIS (build)
container: /root/reference does not exist on the GPU box, so the number is taken here, with the core count
stated, and carried into bench.py's JSON line as `cpu_baseline_reference` (profiles/cpu_reference_baseline.json).

The unmodified reference is imported under tests/golden/ref_shim.py (stub gym, NumPy aliases).  The policy
is the same stand-in as everywhere else -- leaf = mix32(env, t) % (number of valid leaves), computed on the
host from the observation the VecEnv returned (SURVEY.md 8(d) CPU baseline timing (i), (ii)).

    PYTHONDONTWRITEBYTECODE=1 python scripts/time_reference_cpu.py [seconds-per-config] [workers] [only-config ...]
(with config names given, only those are timed and merged into the existing JSON)
"""
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
from tests.golden import ref_shim  # noqa: E402

ref_shim.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gym  # noqa: E402  (the shim's stub)
from tests.common import mix32  # noqa: E402

for _n in ("Dict", "Tuple", "Discrete"):  # wrapper/util.py:38-46 only isinstance-tests against them
    if not hasattr(gym.spaces, _n):
        setattr(gym.spaces, _n, type(_n, (), {}))
if not hasattr(gym.Env, "spec"):
    gym.Env.spec = None  # real gym.Env carries it; wrapper/dummy_vec_env.py:29 reads it
torch.set_num_threads(1)  # main.py:28
gym.register(id="PctDiscrete-v0", entry_point="pct_envs.PctDiscrete0:PackingDiscrete")      # tools.py:232-239
gym.register(id="PctContinuous-v0", entry_point="pct_envs.PctContinuous0:PackingContinuous")
import envs as refenvs  # noqa: E402  (/root/reference/envs.py)
import givenData  # noqa: E402
from wrapper.dummy_vec_env import DummyVecEnv  # noqa: E402

I, L = 80, 50


def args_for(kind, setting, nproc, container=None, bounds=(1.0, 5.0), holders=None):
    cont = kind == "continuous"
    Ih, Lh = holders or (I, L)
    return SimpleNamespace(
        id="PctContinuous-v0" if cont else "PctDiscrete-v0", seed=4, num_processes=nproc, device=torch.device("cpu"),
        setting=setting, container_size=container or givenData.container_size, item_size_set=givenData.item_size_set,
        dataset_path=None, load_dataset=False, internal_node_holder=Ih, leaf_node_holder=Lh, lnes="EMS", shuffle=False,
        sample_from_distribution=cont, sample_left_bound=bounds[0] if cont else None, sample_right_bound=bounds[1] if cont else None)


def drive(venv, n, seconds, warm, I=I, L=L):
    obs = venv.reset()
    g = np.arange(n, dtype=np.uint64)
    t = 0

    def one(obs, t):
        o = obs.view(n, -1, 9).numpy()
        leaf = o[:, I:I + L]
        k = (leaf[:, :, 8] != 0).sum(1).astype(np.uint64)
        h = mix32(g, np.full(n, t, np.uint64))
        idx = np.where(k > 0, h % np.maximum(k, np.uint64(1)), np.uint64(0)).astype(np.int64)
        return venv.step(leaf[np.arange(n), idx])  # float32 [n,9] leaf rows (train_tools.py:66-67)

    for _ in range(warm):
        obs, _, _, _ = one(obs, t)
        t += 1
    t0 = time.perf_counter()
    it = 0
    while time.perf_counter() - t0 < seconds:
        obs, _, _, _ = one(obs, t)
        t += 1
        it += 1
    dt = time.perf_counter() - t0
    return n * it / dt, it, dt


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    only = sys.argv[3:]
    dst = os.path.join(ROOT, "profiles", "cpu_reference_baseline.json")
    out = {"host": {"cores": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__},
           "recipe": "PYTHONDONTWRITEBYTECODE=1 python scripts/time_reference_cpu.py %g %d (build container; the unmodified "
                     "reference under tests/golden/ref_shim.py; torch.set_num_threads(1), OMP_NUM_THREADS=1)" % (seconds, workers),
           "configs": {}}
    if only:
        out = json.load(open(dst))
    cfgs = (("discrete_s2_shmem", "discrete", 2, {}), ("continuous_s2_shmem", "continuous", 2, {}),
            # BASELINE configs[4]'s env (C5): 100^3 bin, 200 / 200 nodes, items U(5, 25)
            ("continuous_c5_shmem", "continuous", 2, dict(container=(100, 100, 100), bounds=(5.0, 25.0), holders=(200, 200))))
    for name, kind, setting, extra in cfgs:
        if only and name not in only:
            continue
        venv = refenvs.make_vec_envs(args_for(kind, setting, workers, **extra), None, True)  # ShmemVecEnv(fork) + VecPyTorch
        Ih, Lh = extra.get("holders", (I, L))
        v, it, dt = drive(venv, workers, seconds, warm=30 if extra else 100, I=Ih, L=Lh)
        venv.close()
        out["configs"][name] = {"value": v, "unit": "env-steps/s", "workers": workers, "iterations": it, "seconds": dt,
                                "path": "envs.make_vec_envs -> ShmemVecEnv(context='fork') + VecPyTorch(cpu)"}
        print(name, "%.1f env-steps/s (%d workers, %d iterations, %.1f s)" % (v, workers, it, dt), flush=True)
    # BASELINE configs[0]: setting 1, one env under DummyVecEnv
    if not only or "discrete_s1_dummy_1env" in only:
        a = args_for("discrete", 1, 1)
        venv = refenvs.VecPyTorch(DummyVecEnv([refenvs.make_env(a.id, a.seed, 0, None, True, a)]), a.device)
        v, it, dt = drive(venv, 1, seconds, warm=100)
        venv.close()
        out["configs"]["discrete_s1_dummy_1env"] = {"value": v, "unit": "env-steps/s", "workers": 1, "iterations": it, "seconds": dt,
                                                    "path": "DummyVecEnv([env]) + VecPyTorch(cpu) (BASELINE configs[0])"}
        print("discrete_s1_dummy_1env %.1f env-steps/s" % v, flush=True)
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
