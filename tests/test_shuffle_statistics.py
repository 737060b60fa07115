"""shuffle=True is the only value main.py can produce (tools.py:136 `type=bool, default=True`).  The batched
env permutes the candidates with a counter-keyed priority and draws its items with a counter-keyed pick where
the reference consumes each worker's NumPy MT19937 stream (bin3D.py:47-54,114-115; binCreator.py:37-39):
bit-exact between the HIP kernels and the oracle, equal IN DISTRIBUTION to the reference.  This file checks
the second half against statistics of the unmodified reference (tests/golden/gen_shuffle_stats.py ->
tests/golden/shuffle_stats.json): mean final utilisation, mean episode length and mean number of valid
leaves of >= 20 000 episodes must agree within 4 standard errors (both sides' sampling error combined)."""
import importlib
import json
import os

import numpy as np
import pytest

from tests.common import GOLDEN, item_set_range

STATS = json.load(open(os.path.join(GOLDEN, "shuffle_stats.json")))["configs"]
CONFIGS = {"discrete_s2_shuffle": (False, 2, 20000), "discrete_s1_shuffle": (False, 1, 6000),
           "continuous_s2_shuffle": (True, 2, 8000)}


def _collect(step, N, episodes, get_obs, I=80, L=50):
    ratios, lengths, leaves = [], [], []
    while len(ratios) < episodes:
        o = get_obs()
        leaves.append(float((o.reshape(N, -1, 9)[:, I:I + L, 8] != 0).sum(1).mean()))
        done, counter, ratio = step()
        d = np.nonzero(done)[0]
        ratios.extend(ratio[d].tolist())
        lengths.extend(counter[d].tolist())
    return np.asarray(ratios), np.asarray(lengths, np.float64), float(np.mean(leaves))


def _check(name, ratios, lengths, leaves):
    ref = STATS[name]
    for what, ours, mean, var in (("ratio", ratios, ref["ratio_mean"], ref["ratio_var"]),
                                  ("length", lengths, ref["length_mean"], ref["length_var"])):
        se = np.sqrt(var / ref["episodes"] + ours.var() / len(ours))
        assert abs(ours.mean() - mean) < 4 * se, (name, what, ours.mean(), mean, se)
    # valid leaves per observation: a mean over correlated steps -- a loose band is enough to catch a broken cut
    assert abs(leaves - ref["valid_leaves_mean"]) < 0.05 * ref["valid_leaves_mean"], (leaves, ref["valid_leaves_mean"])


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_shuffle_statistics_match_reference(name):
    from oracle.oracle_lib import OracleVecEnv
    cont, setting, episodes = CONFIGS[name]
    episodes //= 4  # the CPU leg keeps to a few seconds
    N = 256
    if cont:
        ora = OracleVecEnv(N, setting=setting, container_size=(10, 10, 10), env_kind=1, sample_bounds=(1.0, 5.0),
                           shuffle=True, shuffle_seed=4, threads=min(os.cpu_count() or 1, 16))
    else:
        ora = OracleVecEnv(N, setting=setting, container_size=(10, 10, 10), item_set=item_set_range(1, 5), shuffle=True,
                           shuffle_seed=4, threads=min(os.cpu_count() or 1, 16))
    ora.set_sampler(4)
    ora.reset()

    def step():
        ora.step_hash_policy(1)
        return ora.done.astype(bool), ora.counter.copy(), ora.ratio.copy()

    _check(name, *_collect(step, N, episodes, lambda: ora.obs))
    ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_hip_shuffle_statistics_match_reference(name):
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    cont, setting, episodes = CONFIGS[name]
    N = 1024
    if cont:
        env = pkg.PctVecEnv(N, setting=setting, container_size=(10, 10, 10), continuous=True, sample_left_bound=1.0,
                            sample_right_bound=5.0, shuffle=True, seed=4, device="cuda:0")
    else:
        env = pkg.PctVecEnv(N, setting=setting, container_size=(10, 10, 10), item_set=item_set_range(1, 5), shuffle=True,
                            seed=4, device="cuda:0")
    env.reset()

    def step():
        env.step_hash_policy(1)
        _, _, done, _ = env.step_wait()
        return done, env._h_counter.numpy().copy(), env._h_ratio.numpy().copy()

    _check(name, *_collect(step, N, episodes, lambda: env.current_obs().cpu().numpy()))
    assert not env.error_flags.any()
    env.close()
