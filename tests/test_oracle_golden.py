"""CPU: the oracle (oracle/pct_oracle.c) against the committed reference fixtures.

The fixtures were produced by tests/golden/gen_golden.py from the UNMODIFIED Python
reference; they are what pins the oracle (and through it the HIP path) to the reference on
a box where /root/reference does not exist."""
import hashlib
import random

import numpy as np
import pytest

from oracle.oracle_lib import OracleVecEnv, pyset_order
from tests.common import case_items, case_density, LNES_CODE, CONT_CASES, DATASET_CASES, GOLDEN_CASES, ORACLE_ONLY_CASES, dataset_trajectories, gather_rows, hash_policy_index, item_set_range, load_case, GOLDEN


@pytest.mark.parametrize("name", GOLDEN_CASES + ORACLE_ONLY_CASES)
def test_oracle_matches_reference_fixture_fused_policy(name):
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"],
                       item_set=case_items(c), internal_node_holder=c["I"],
                       leaf_node_holder=c["L"], env_id_base=c["base"], lnes=LNES_CODE[c.get("lnes", "EMS")])
    env.set_item_stream(z["stream"])
    if case_density(z) is not None:
        env.set_density_stream(case_density(z))
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        assert np.array_equal(env.reward, z["reward"][t])
        assert np.array_equal(env.done, z["done"][t])
        assert np.array_equal(env.counter, z["counter"][t])
        assert np.array_equal(env.ratio * (env.done != 0), z["ratio"][t])
    assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
    assert not env.flags.any()


@pytest.mark.parametrize("mode", ["rows9", "rows6", "index"])
def test_oracle_action_forms_agree(mode):
    """9-vector rows (trainer), 6-vector rows (evaluation_tools.py:24) and leaf indices all
    reproduce the fixture."""
    name = "discrete_s2_rect_60_30"
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"],
                       item_set=case_items(c), internal_node_holder=c["I"],
                       leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    env.reset()
    for t in range(c["steps"]):
        obs32 = env.obs.astype(np.float32)
        assert np.array_equal(obs32, z["obs"][t])
        idx = hash_policy_index(obs32, c["I"], c["L"], c["base"], np.full(c["N"], t, np.uint64))
        if mode == "index":
            env.step_index(idx)
        else:
            rows = gather_rows(obs32, c["I"], idx)
            env.step_rows(rows[:, :6] if mode == "rows6" else rows)
        assert np.array_equal(env.done, z["done"][t])
        assert np.array_equal(env.reward, z["reward"][t])


def test_oracle_three_vector_action_form():
    """(flag, lx, ly) heuristic form (bin3D.py:152-153): flag swaps x and y of the raw item."""
    items = np.array([[[2, 3, 4], [5, 1, 2], [1, 1, 1]]], np.int32)
    env = OracleVecEnv(1, setting=2, container_size=(10, 10, 10), item_set=item_set_range(1, 5))
    env.set_item_stream(items)
    env.reset()
    env.step_rows(np.array([[1.0, 0.0, 0.0]]))  # rotated: x=3, y=2
    row0 = env.obs[0].reshape(-1, 9)[0]
    assert list(row0[:6]) == [0, 0, 0, 3, 2, 4] and row0[6] == 1 and row0[8] == 1
    assert env.reward[0] == pytest.approx(10 * 24 / 1000) and env.done[0] == 0 and env.counter[0] == 1
    env.step_rows(np.array([[0.0, 8.0, 8.0]]))  # 5 wide at x=8 leaves the bin -> episode ends
    assert env.done[0] == 1 and env.reward[0] == 0 and env.counter[0] == 1
    assert env.ratio[0] == pytest.approx(24 / 1000)


def test_known_answer_hash_replay():
    """sha256[:16] of 500 float32 observations of the reference under
    env.seed(4) / RandomState(0) policy = e882162eebfb9734; the recorded item draws and
    actions are replayed through the oracle."""
    z = np.load(GOLDEN + "/kat_discrete_s2.npz")
    env = OracleVecEnv(1, setting=2, container_size=(10, 10, 10), item_set=item_set_range(1, 5))
    env.set_item_stream(z["items"][None])
    env.reset()
    h = hashlib.sha256()
    for t in range(500):
        h.update(env.obs[0].astype(np.float32).tobytes())
        env.step_rows(z["actions"][t][None].astype(np.float64))
    assert h.hexdigest()[:16] == str(z["sha256_16"]) == "e882162eebfb9734"


def test_pyset_order_matches_this_cpython():
    """The emulated set iteration order against the interpreter's own `set` (CPython 3.10
    is what the reference was probed under; the algorithm is unchanged 3.7-3.12)."""
    rnd = random.Random(5)
    for trial in range(300):
        n = rnd.choice([1, 4, 6, 19, 20, 77, 78, 300, 310, 1229, 1500])
        hi = rnd.randint(2, 12)
        keys = [tuple(rnd.randint(0, hi) for _ in range(6)) for _ in range(n)]
        s = set()
        for k in keys:
            s.add(k)
        order = pyset_order(np.array(keys, np.int64))
        assert [keys[i] for i in order] == list(s)


def test_reset_specific_and_sampler_determinism():
    a = OracleVecEnv(6, item_set=item_set_range(1, 5), env_id_base=10)
    b = OracleVecEnv(6, item_set=item_set_range(1, 5), env_id_base=10)
    a.set_sampler(99)
    b.set_sampler(99)
    a.reset()
    b.reset()
    for _ in range(40):
        a.step_hash_policy(1)
        b.step_hash_policy(1)
    assert np.array_equal(a.obs, b.obs)
    before = a.obs.copy()
    a.reset(env_ids=[1, 4])
    changed = [e for e in range(6) if not np.array_equal(before[e], a.obs[e])]
    assert set(changed) <= {1, 4}
    for e in (1, 4):
        st = a.debug_state(e)
        assert st["n_boxes"] == 0 and len(st["ems"]) == 1 and st["heightmap"].sum() == 0


def test_edge_cases_zero_row_and_bad_action():
    env = OracleVecEnv(2, item_set=item_set_range(1, 5))
    env.set_item_stream(np.array([[[5, 5, 5]], [[2, 3, 4]]], np.int32))
    env.reset()
    # env 0: all-zero row -> (0,0,0) + unrotated item (bin3D.py:140): succeeds in an empty bin
    # env 1: extents that are not a permutation of the item -> ValueError in the reference
    rows = np.zeros((2, 9))
    rows[1, :6] = [0, 0, 0, 7, 7, 10]
    env.step_rows(rows)
    assert env.done[0] == 0 and env.counter[0] == 1
    assert env.done[1] == 1 and env.flags[1] == 8


@pytest.mark.parametrize("name", CONT_CASES)
def test_continuous_oracle_matches_reference_fixture(name):
    """PctContinuous0, setting 2: float64 observations bit-exact against the reference fixture
    (same float64 operation order, same CPython set order over float tuples)."""
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], env_kind=1,
                       sample_bounds=(c["lo"], c["hi"]), internal_node_holder=c["I"], leaf_node_holder=c["L"],
                       env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    if case_density(z) is not None:
        env.set_density_stream(case_density(z))
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs, z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        assert np.array_equal(env.reward, z["reward"][t])
        assert np.array_equal(env.done, z["done"][t])
        assert np.array_equal(env.counter, z["counter"][t])
        assert np.array_equal(env.ratio * (env.done != 0), z["ratio"][t])
    assert np.array_equal(env.obs, z["obs"][c["steps"]])
    assert not env.flags.any()


def test_continuous_known_answer_hash_replay():
    """506b5c0349c89b9d (observations rounded to 5 decimals, float32)."""
    z = np.load(GOLDEN + "/kat_continuous_s2.npz")
    env = OracleVecEnv(1, setting=2, container_size=(10, 10, 10), env_kind=1, sample_bounds=(1.0, 5.0))
    env.set_item_stream(z["items"][None])
    env.reset()
    h = hashlib.sha256()
    for t in range(500):
        h.update(np.round(env.obs[0], 5).astype(np.float32).tobytes())
        env.step_rows(z["actions"][t][None])
    assert h.hexdigest()[:16] == str(z["sha256_16"]) == "506b5c0349c89b9d"


def test_known_answer_hash_replay_setting1():
    """discrete setting 1 (stability check) = 443198ae2c0162db."""
    z = np.load(GOLDEN + "/kat_discrete_s1.npz")
    env = OracleVecEnv(1, setting=1, container_size=(10, 10, 10), item_set=item_set_range(1, 5))
    env.set_item_stream(z["items"][None])
    env.reset()
    h = hashlib.sha256()
    for t in range(500):
        h.update(env.obs[0].astype(np.float32).tobytes())
        env.step_rows(z["actions"][t][None].astype(np.float64))
    assert h.hexdigest()[:16] == str(z["sha256_16"]) == "443198ae2c0162db"


@pytest.mark.parametrize("name", DATASET_CASES)
def test_oracle_dataset_semantics_match_reference(name):
    """LoadBoxCreator (binCreator.py:41-72): episode k plays trajectory k (1-based), sentinel
    (100,100,100) after the last item, then (10,10,10)."""
    c, z = load_case(name)
    trajs = dataset_trajectories(z)
    dens = [t[:, 3] for t in trajs] if c["setting"] == 3 else None
    if c["kind"] == "discrete":
        env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                           internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_dataset([t[:, :3].astype(np.int32) for t in trajs], dens)
    else:
        env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], env_kind=1, sample_bounds=(c["lo"], c["hi"]),
                           internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_dataset([np.rint(t[:, :3] * 1000).astype(np.int32) for t in trajs], dens)
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(z["obs"].dtype), z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t])


def test_shuffle_is_a_permutation_of_the_unshuffled_candidates():
    """shuffle=True (bin3D.py:114-115): with L large enough to hold every feasible candidate the
    shuffled leaf rows are a permutation of the unshuffled ones; with the default L the first-L
    cut differs, and the permutation depends on the seed."""
    items = item_set_range(1, 5)
    stream = np.array([[[3, 4, 2], [2, 2, 5], [4, 1, 3], [5, 5, 1], [2, 3, 3], [1, 4, 4]]], np.int32)
    envs = []
    for sh, seed in ((False, 0), (True, 1), (True, 2)):
        e = OracleVecEnv(1, item_set=items, leaf_node_holder=600, shuffle=sh, shuffle_seed=seed)
        e.set_item_stream(stream)
        e.reset()
        envs.append(e)
    for t in range(5):
        rows = [sorted(map(tuple, e.obs[0].reshape(-1, 9)[80:680][e.obs[0].reshape(-1, 9)[80:680, 8] != 0])) for e in envs]
        assert rows[0] == rows[1] == rows[2]
        seqs = [e.obs[0].reshape(-1, 9)[80:680].copy() for e in envs]
        if len(rows[0]) > 3:
            assert not np.array_equal(seqs[0], seqs[1]) and not np.array_equal(seqs[1], seqs[2])
        first = seqs[0][0].copy()  # same placement everywhere keeps the states identical
        for e in envs:
            e.step_rows(first[None])


@pytest.mark.parametrize("name,heur", [(n, h) for n in ["heur_s2_10", "heur_s1_10", "heur_s2_rect"]
                                       for h in ["LSAH", "HM", "OnlineBPH", "DBL", "BR", "RANDOM"]] +
                         [("heur_macs_s2_10", "MACS"), ("heur_macs_s1_rect", "MACS")])
def test_oracle_heuristics_match_reference_loops(name, heur):
    """heuristic.py's loops (LASH, heightmap_min, OnlineBPH, DBL, BR) run on the unmodified reference:
    per-episode utilisation and number of packed items of the same item stream."""
    from tests.common import HEUR_CODE
    c, z = load_case(name)
    env = OracleVecEnv(1, setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"])
    env.set_item_stream(z["stream"])
    env.reset()
    util, length = [], []
    while len(util) < c["episodes"]:
        env.step_heuristic(HEUR_CODE[heur], 1)
        if env.done[0]:
            util.append(float(env.ratio[0]))
            length.append(int(env.counter[0]))
    assert np.array_equal(np.array(util), z["util_" + heur])
    assert np.array_equal(np.array(length, np.int32), z["len_" + heur])
    assert not env.flags.any()


@pytest.mark.parametrize("name", ["heur_cont_s2_10", "heur_cont_s1_unit"])
@pytest.mark.parametrize("heur", ["LSAH", "OnlineBPH", "BR"])
def test_oracle_continuous_heuristics_match_reference_loops(name, heur):
    """heuristic.py's LASH / OnlineBPH / BR run on the unmodified PackingContinuous (the three tools.py:217-218 allows
    there): per-episode utilisation and number of packed items of the same item stream."""
    from tests.common import HEUR_CODE
    c, z = load_case(name)
    env = OracleVecEnv(1, setting=c["setting"], container_size=c["container"], env_kind=1, item_set=[(c["lo"], c["lo"], c["lo"])],
                       internal_node_holder=c["I"], leaf_node_holder=c["L"])
    env.set_item_stream(z["stream"])
    env.reset()
    util, length = [], []
    while len(util) < c["episodes"]:
        env.step_heuristic(HEUR_CODE[heur], 1)
        if env.done[0]:
            util.append(float(env.ratio[0]))
            length.append(int(env.counter[0]))
    assert np.array_equal(np.array(util), z["util_" + heur])
    assert np.array_equal(np.array(length, np.int32), z["len_" + heur])
    assert not env.flags.any()
    env.close()


def test_oracle_notice_precedes_the_lapack_divergence():
    """tests/golden/discrete_s1_flat_diverging.npz is the UNMODIFIED reference on the one adversarial stream (seed 66) whose
    env 0 meets a least-squares system with a rank decision at the cut (sigma_max 1.1e15, smallest kept singular value
    1.97 x the rcond cut): from step 79 on the reference's trajectory is LAPACK dgelsd's.  The oracle must equal the
    reference before that step and on every other env throughout, and its notice (the product's
    PCT_FLAG_ILL_CONDITIONED) must be up on env 0 no later than that step and on no other env.  This is the JACOBI stand-in
    (rounds 1-4's default, now opt-in); in the default mode the oracle follows the recording to its end (tests/test_gelsd_port.py)."""
    from oracle import oracle_lib
    old = oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_JACOBI)
    try:
        _notice_precedes_the_divergence()
    finally:
        oracle_lib.set_lstsq_mode(old)


def _notice_precedes_the_divergence():
    c, z = load_case("discrete_s1_flat_diverging")
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    env.reset()
    div = z["first_divergence"]
    assert list(div) == [79, -1, -1, -1]
    alive = np.ones(c["N"], bool)
    for t in range(c["steps"]):
        bad = (env.obs.astype(np.float32) != z["obs"][t]).any(1)
        ill = env.ill_conditioned()
        for e in range(c["N"]):
            if alive[e] and bad[e]:
                assert t == div[e] and ill[e], (e, t)  # parts ways exactly where recorded, with the notice already up
                alive[e] = False
        env.step_hash_policy(1)
    assert list(alive) == [False, True, True, True]
    assert list(env.ill_conditioned()) == [True, False, False, False]
    env.close()


def test_oracle_matches_reference_plate_fixture():
    """tests/golden/gen_plate_golden.py: the unmodified reference under scripted 3-vector actions -- unit tiles, a plate on 20 / 22 / 24
    of them none of which holds its centre of mass (np.linalg.lstsq over that many unknowns, the widest systems dgelsd still solves on
    the restated path: n <= SMLSIZ = 25), boxes on the plate.  Every observation, reward, done, counter, ratio."""
    c, z = load_case("plate_discrete_s1")
    items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=items, internal_node_holder=c["I"],
                       leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    env.reset()
    assert max(c["lstsq_widths"]) == 24 and c["lstsq_widths"][24] >= 20
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), t
        env.step_rows(z["actions"][t].astype(np.float64))
        assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t]), t
        assert np.array_equal(env.counter, z["counter"][t]) and np.array_equal(env.ratio * (env.done != 0), z["ratio"][t]), t
    assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
    assert not env.flags.any()
    env.close()

