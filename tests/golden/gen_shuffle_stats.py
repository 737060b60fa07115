"""Reference statistics for the CLI's default configuration -- shuffle=True (tools.py:136) and on-the-fly
RandomBoxCreator items (binCreator.py:37-39), every env seeded `seed + rank` as envs.py:49 does under
ShmemVecEnv(fork) -- which the batched env reproduces in DISTRIBUTION, not draw for draw (its item picks and
its candidate shuffle are counter-keyed, not NumPy's MT19937 stream).

Run in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_shuffle_stats.py [episodes-per-config]

Per config: the unmodified reference env, `np.random.seed(seed + rank)` before each env's run (one process per
env in the reference, so each env owns its stream), the stand-in hash policy (leaf = mix32(rank, t) % valid
leaves -- which leaf that is depends on the shuffled order), until `episodes` episodes have finished.  Written:
tests/golden/shuffle_stats.json with per-config episode count, mean / variance of the final space utilisation
and of the episode length (packed boxes), and mean number of valid leaves per observation.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import ref_shim  # noqa: E402
from gen_golden import mix32  # noqa: E402

PackingDiscrete, PackingContinuous, item_size_set = ref_shim.load_reference_envs()
EPISODES = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
I, L = 80, 50


def run(kind, setting, episodes, envs=32, seed=4):
    ratios, lengths, leaves = [], [], []
    per_env = (episodes + envs - 1) // envs
    for rank in range(envs):
        np.random.seed(seed + rank)  # bin3D.py:47-54 via envs.py:49
        if kind == "discrete":
            env = PackingDiscrete(setting=setting, container_size=[10, 10, 10], item_set=item_size_set,
                                  internal_node_holder=I, leaf_node_holder=L, LNES="EMS", shuffle=True)
        else:
            env = PackingContinuous(setting=setting, container_size=[10, 10, 10], item_set=item_size_set,
                                    internal_node_holder=I, leaf_node_holder=L, LNES="EMS", shuffle=True,
                                    sample_from_distribution=True, sample_left_bound=1.0, sample_right_bound=5.0)
        obs = env.reset()
        done_eps, t = 0, 0
        while done_eps < per_env:
            leaf = obs.reshape(-1, 9)[I:I + L]
            k = int((leaf[:, 8] != 0).sum())
            leaves.append(k)
            a = leaf[mix32(rank, t) % k] if k > 0 else leaf[0]
            obs, r, done, info = env.step(a.copy())
            t += 1
            if done:
                ratios.append(info["ratio"])
                lengths.append(info["counter"])
                done_eps += 1
                obs = env.reset()
    ratios, lengths = np.asarray(ratios), np.asarray(lengths, np.float64)
    return {"episodes": int(len(ratios)), "ratio_mean": float(ratios.mean()), "ratio_var": float(ratios.var()),
            "length_mean": float(lengths.mean()), "length_var": float(lengths.var()),
            "valid_leaves_mean": float(np.mean(leaves)), "envs": envs, "seed": seed}


out = {"recipe": "PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_shuffle_stats.py %d" % EPISODES, "configs": {}}
for name, kind, setting, eps in (("discrete_s2_shuffle", "discrete", 2, EPISODES), ("discrete_s1_shuffle", "discrete", 1, EPISODES // 5),
                                 ("continuous_s2_shuffle", "continuous", 2, EPISODES // 4)):
    out["configs"][name] = run(kind, setting, eps)
    print(name, out["configs"][name], flush=True)
json.dump(out, open(os.path.join(HERE, "shuffle_stats.json"), "w"), indent=1)
