"""Strict NumPy-stream mode: with the CLI's own defaults -- shuffle=True (tools.py:136), on-the-fly RandomBoxCreator
items (binCreator.py:37-39), setting-3 densities from np.random.random (bin3D.py:82-84) -- and every env's
RandomState seeded `seed + rank` (envs.py:49, bin3D.py:47-54) the reference's trajectory is a function of the seed
and the actions alone.  The fixtures `discrete_s*_numpy_stream.npz` are runs of the UNMODIFIED reference in exactly
that configuration (tests/golden/gen_golden.py numpy_stream_cases: nothing scripted, nothing patched); the oracle's
and the HIP kernels' NumPy-stream mode (per-env MT19937 state, legacy randint / random_sample / shuffle consumption
order, the extra draws of a failed step's discarded observation) must reproduce them bit for bit."""
import importlib

import numpy as np
import pytest

from tests.common import hash_policy_index, item_set_range, load_case

CASES = ["discrete_s2_numpy_stream", "discrete_s1_numpy_stream", "discrete_s3_numpy_stream",
         # bin3D.py:114-115 shuffles whatever --lnes produced, in any bin: the other leaf expansions and bins beyond 31
         # cells per axis (64-bit candidate keys in the kernels), again against unscripted reference runs
         "discrete_s2_numpy_stream_cp", "discrete_s1_numpy_stream_cp", "discrete_s2_numpy_stream_ep",
         "discrete_s3_numpy_stream_ep", "discrete_s2_numpy_stream_ev", "discrete_s2_numpy_stream_fc",
         "discrete_s1_numpy_stream_fc", "discrete_s2_numpy_stream_u64", "discrete_s1_numpy_stream_u64",
         "discrete_s2_numpy_stream_u64_cp",
         # ... and the stability settings under each of them
         "discrete_s3_numpy_stream_fc", "discrete_s1_numpy_stream_ep", "discrete_s1_numpy_stream_ev",
         "discrete_s3_numpy_stream_cp", "discrete_s3_numpy_stream_u64", "discrete_s1_numpy_stream_u64_ep"]
LNES_ID = {"EV": 1, "EP": 2, "CP": 3, "FC": 4}
# continuous env in its sampling mode (C/bin3D.py:14-16,103-113): items are round(np.random.uniform(a, b), 3), plus
# the RandomBoxCreator's unread randint, the density and the shuffle; a failed step's discarded observation draws a
# whole new item
CONT_CASES = ["continuous_s2_numpy_stream", "continuous_s1_numpy_stream", "continuous_s3_numpy_stream"]
GIVEN_ITEM_COUNT = 125  # len(givenData.item_size_set)


def test_mt19937_restatement_matches_numpy():
    """the oracle's draws (through a 1-item-set env would be indirect): the same init_genrand / tempering / masked
    rejection restated here must equal np.random's legacy stream"""
    for seed in (0, 4, 123456789):
        rs = np.random.RandomState(seed)
        mt = np.zeros(624, np.uint64)
        s = seed
        for i in range(624):
            mt[i] = s
            s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
        assert np.array_equal(rs.get_state()[1].astype(np.uint64), mt)
        # randint(0, 125) = masked rejection on 32-bit words with mask 127; shuffle = random_interval from the back
        words = rs.randint(0, 2 ** 32, size=4000, dtype=np.uint64)
        rs2 = np.random.RandomState(seed)
        it, got = iter(words), []
        for _ in range(300):
            while True:
                v = int(next(it)) & 127
                if v <= 124:
                    break
            got.append(v)
        assert got == [int(rs2.randint(0, 125)) for _ in range(300)]


@pytest.mark.parametrize("name", CASES)
def test_oracle_numpy_stream_matches_reference(name):
    from oracle.oracle_lib import OracleVecEnv
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=item_set_range(c["lo"], c["hi"]),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"], shuffle=True,
                       lnes=LNES_ID.get(c.get("lnes"), 0))
    env.set_numpy_rng(c["seed"])
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t]), (name, t)
        assert np.array_equal(env.counter, z["counter"][t])
    assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
    env.close()


@pytest.mark.parametrize("name", CONT_CASES)
def test_oracle_numpy_stream_continuous_matches_reference(name):
    from oracle.oracle_lib import OracleVecEnv
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], env_kind=1, sample_bounds=(c["lo"], c["hi"]),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"], shuffle=True)
    env.set_numpy_rng(c["seed"], n_item_set=GIVEN_ITEM_COUNT)
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs, z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t]), (name, t)
        assert np.array_equal(env.counter, z["counter"][t])
        assert np.array_equal(env.ratio * (env.done != 0), z["ratio"][t])
    assert np.array_equal(env.obs, z["obs"][c["steps"]])
    env.close()


def test_round3_restatement_matches_python_round():
    """the item edges are Python round(x, 3) of a uniform draw: correctly rounded on the binary value, ties to even.
    The oracle's / kernel's integer midpoint comparison restated here must agree with round() -- also on the doubles
    right next to a decimal midpoint, where x * 1000 in floating point rounds the wrong way."""
    from fractions import Fraction

    def round3_lattice(x):
        k = int(x * 1000.0 + 0.5)
        for _ in range(3):
            fx = Fraction(x) * 2000
            up, dn = fx - (2 * k + 1), fx - (2 * k - 1)
            if up > 0 or (up == 0 and k & 1):
                k += 1
                continue
            if dn < 0 or (dn == 0 and k & 1):
                k -= 1
                continue
            break
        return k

    rs = np.random.RandomState(5)
    xs = list(0.1 + 0.4 * rs.random_sample(2000)) + list(1.0 + 4.0 * rs.random_sample(2000)) + list(5.0 + 20.0 * rs.random_sample(2000))
    for k in range(100, 5000, 37):
        mid = (2 * k + 1) / 2000.0
        xs += [mid, np.nextafter(mid, 0.0), np.nextafter(mid, 10.0)]
    for x in xs:
        x = float(x)
        assert round3_lattice(x) / 1000.0 == round(x, 3), x


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONT_CASES)
@pytest.mark.parametrize("mode", ["fused", "rows9"])
def test_hip_numpy_stream_continuous_matches_reference(name, mode):
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    c, z = load_case(name)
    env = pkg.PctVecEnv(c["N"], setting=c["setting"], container_size=c["container"], continuous=True,
                        sample_from_distribution=True, sample_left_bound=c["lo"], sample_right_bound=c["hi"],
                        item_set=np.zeros((GIVEN_ITEM_COUNT, 3)), internal_node_holder=c["I"], leaf_node_holder=c["L"],
                        env_id_base=c["base"], shuffle=True, rng="numpy", seed=c["seed"], device="cuda:0")
    obs = env.reset()
    for t in range(c["steps"]):
        o = obs.cpu().numpy()
        want = z["obs"][t].astype(np.float32)
        assert np.array_equal(o, want), (name, mode, t, np.argwhere(o != want)[:4])
        if mode == "fused":
            env.step_hash_policy(1)
            obs, reward, done, infos = env.step_wait()
        else:
            idx = hash_policy_index(o, c["I"], c["L"], c["base"], np.full(c["N"], t, np.uint64))
            rows = o.reshape(c["N"], -1, 9)[np.arange(c["N"]), c["I"] + idx].copy()
            obs, reward, done, infos = env.step(rows)
        assert np.array_equal(done.astype(np.uint8), z["done"][t]), (name, t)
        assert np.array_equal(reward[:, 0].numpy(), z["reward"][t].astype(np.float32)), (name, t)
        for i in np.nonzero(done)[0]:
            assert infos[i]["ratio"] == z["ratio"][t][i]
    assert not env.error_flags.any()
    env.close()


@pytest.mark.gpu
def test_hip_numpy_stream_continuous_batched_vs_oracle():
    """many envs and resets, every setting, with the small-capacity launch + retry pass in play"""
    from oracle.oracle_lib import OracleVecEnv
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    N = 256
    for setting, cont, lo, hi in ((2, (10, 10, 10), 1.0, 5.0), (1, (1, 1, 1), 0.1, 0.5), (3, (1, 1, 1), 0.1, 0.5)):
        kw = dict(setting=setting, container_size=cont, internal_node_holder=80, leaf_node_holder=50, env_id_base=300, shuffle=True)
        env = pkg.PctVecEnv(N, continuous=True, sample_from_distribution=True, sample_left_bound=lo, sample_right_bound=hi,
                            rng="numpy", seed=91, device="cuda:0", **kw)
        ora = OracleVecEnv(N, threads=16, env_kind=1, sample_bounds=(lo, hi), **kw)
        ora.set_numpy_rng(91)
        obs = env.reset()
        ora.reset()
        for t in range(120):
            if t % 10 == 0:
                assert np.array_equal(obs.cpu().numpy(), ora.obs.astype(np.float32)), (setting, t)
            env.step_hash_policy(1)
            ora.step_hash_policy(1)
            obs, reward, done, infos = env.step_wait()
            assert np.array_equal(done.astype(np.uint8), ora.done), (setting, t)
            assert np.array_equal(reward[:, 0].numpy(), ora.reward.astype(np.float32))
        assert not env.error_flags.any()
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["fused", "rows9"])
def test_hip_numpy_stream_matches_reference(name, mode):
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    c, z = load_case(name)
    env = pkg.PctVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=item_set_range(c["lo"], c["hi"]),
                        internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"], shuffle=True,
                        rng="numpy", seed=c["seed"], device="cuda:0", LNES=c.get("lnes", "EMS"))
    obs = env.reset()
    for t in range(c["steps"]):
        o = obs.cpu().numpy()
        assert np.array_equal(o, z["obs"][t]), (name, mode, t, np.argwhere(o != z["obs"][t])[:4])
        if mode == "fused":
            env.step_hash_policy(1)
            obs, reward, done, infos = env.step_wait()
        else:
            idx = hash_policy_index(o, c["I"], c["L"], c["base"], np.full(c["N"], t, np.uint64))
            rows = o.reshape(c["N"], -1, 9)[np.arange(c["N"]), c["I"] + idx].copy()
            obs, reward, done, infos = env.step(rows)
        assert np.array_equal(done.astype(np.uint8), z["done"][t]), (name, t)
        assert np.array_equal(reward[:, 0].numpy(), z["reward"][t].astype(np.float32)), (name, t)
        for i in np.nonzero(done)[0]:
            assert infos[i]["ratio"] == z["ratio"][t][i]
    assert not env.error_flags.any()
    env.close()


@pytest.mark.gpu
def test_hip_numpy_stream_batched_vs_oracle():
    """many envs, many resets, long enough for several regenerations of every env's 624-word block"""
    from oracle.oracle_lib import OracleVecEnv
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    N, items = 512, item_set_range(1, 5)
    for setting in (2, 3):
        kw = dict(setting=setting, container_size=(10, 10, 10), item_set=items, internal_node_holder=80, leaf_node_holder=50,
                  env_id_base=1000, shuffle=True)
        env = pkg.PctVecEnv(N, rng="numpy", seed=77, device="cuda:0", **kw)
        ora = OracleVecEnv(N, threads=16, **kw)
        ora.set_numpy_rng(77)
        obs = env.reset()
        ora.reset()
        for t in range(150):
            if t % 10 == 0:
                assert np.array_equal(obs.cpu().numpy(), ora.obs.astype(np.float32)), (setting, t)
            env.step_hash_policy(1)
            ora.step_hash_policy(1)
            obs, reward, done, infos = env.step_wait()
            assert np.array_equal(done.astype(np.uint8), ora.done), (setting, t)
            assert np.array_equal(reward[:, 0].numpy(), ora.reward.astype(np.float32))
        assert not env.error_flags.any()
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["discrete", "continuous"])
def test_hip_numpy_stream_with_heavy_first_dispatch(kind, monkeypatch):
    """strict mode under the heavy-first dispatch (forced on: PCT_ORDER=1): the env's MT19937 state travels with the env id,
    not with the workgroup that happens to step it; multi-step launches included"""
    from oracle.oracle_lib import OracleVecEnv
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    monkeypatch.setenv("PCT_EXPERIMENT", "1")  # (the knobs are read only under it)
    monkeypatch.setenv("PCT_ORDER", "1")
    N = 384
    if kind == "discrete":
        kw = dict(setting=3, container_size=(10, 10, 10), item_set=item_set_range(1, 5), internal_node_holder=80,
                  leaf_node_holder=50, env_id_base=9, shuffle=True)
        env = pkg.PctVecEnv(N, rng="numpy", seed=21, device="cuda:0", **kw)
        ora = OracleVecEnv(N, threads=16, **kw)
        ora.set_numpy_rng(21)
    else:
        kw = dict(setting=2, container_size=(10, 10, 10), internal_node_holder=80, leaf_node_holder=50, env_id_base=9, shuffle=True)
        env = pkg.PctVecEnv(N, rng="numpy", seed=21, device="cuda:0", continuous=True, sample_left_bound=1.0,
                            sample_right_bound=5.0, **kw)
        ora = OracleVecEnv(N, threads=16, env_kind=1, sample_bounds=(1.0, 5.0), **kw)
        ora.set_numpy_rng(21, n_item_set=GIVEN_ITEM_COUNT)
    obs = env.reset()
    ora.reset()
    for t in range(40):
        n = 3 if t % 7 == 6 else 1
        env.step_hash_policy(n)
        ora.step_hash_policy(n)
        obs, reward, done, infos = env.step_wait()
        assert np.array_equal(done.astype(np.uint8), ora.done), (kind, t)
        assert np.array_equal(obs.cpu().numpy(), ora.obs.astype(np.float32)), (kind, t)
    assert not env.error_flags.any()
    env.close()
    ora.close()
