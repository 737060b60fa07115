"""GPU: pct_set_lstsq_mode(PCT_LSTSQ_GELSD) -- the kernels' strict least-squares solver (csrc/pct_gelsd.cuh: LAPACK dgelsd as the
reference's NumPy executes it) against the unmodified reference's fixtures and against the oracle's independent restatement
(oracle/pct_oracle_gelsd.c) on the same seeded inputs.  Bit-exact, through the C ABI via PctVecEnv(lstsq="gelsd").

(The file sorts last on purpose: these are the round-4 additions; the suites of the default solver run before them.)"""
import importlib
import os

import numpy as np
import pytest

from oracle import oracle_lib
from oracle.oracle_lib import OracleVecEnv
from tests.common import case_density, case_items, load_case, make_stream

pytestmark = pytest.mark.gpu


def _pkg():
    return importlib.import_module("online-3d-bpp-pct_amd")


@pytest.fixture
def oracle_gelsd():
    old = oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_GELSD)
    yield
    oracle_lib.set_lstsq_mode(old)


@pytest.mark.parametrize("name", ["discrete_s1_flat_diverging", "discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq", "discrete_s1_flat20",
                                  "discrete_s1_10_80_50"])
def test_hip_gelsd_matches_reference_fixture(name):
    """the unmodified reference's recordings: every observation of every env.  discrete_s1_flat_diverging is the stream on
    which the default solver leaves the reference at step 79 (tests/test_gpu_parity.py::test_hip_notice_precedes_...): in
    gelsd mode the kernels follow the reference through that rank decision to the end of the recording."""
    c, z = load_case(name)
    env = _pkg().PctVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                           internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"], item_stream=z["stream"],
                           device="cuda:0", lstsq="gelsd")
    if case_density(z) is not None:
        env.set_density_stream(case_density(z))
    obs = env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(obs.cpu().numpy(), z["obs"][t]), (name, t)
        env.step_hash_policy(1)
        obs, reward, done, _ = env.step_wait()
        assert np.array_equal(done.astype(np.uint8), z["done"][t]), (name, t)
    assert np.array_equal(obs.cpu().numpy(), z["obs"][c["steps"]])
    assert not env.error_flags.any()
    if name == "discrete_s1_flat_diverging":
        assert env.ill_conditioned[0]  # the notice is still raised at the rank cut
    env.close()


@pytest.mark.parametrize("N_pad", [0, 60])
def test_hip_matches_reference_plate_fixture(N_pad):
    """Splits over MORE THAN 16 supporters (round 6): the unmodified reference under scripted 3-vector actions -- unit tiles, a plate
    on 20 / 22 / 24 of them none of which holds its centre of mass, boxes on the plate (tests/golden/gen_plate_golden.py: 154
    np.linalg.lstsq calls over 20 - 24 unknowns).  Such a split is a capacity of the normal pass (8 supporters) AND of round 5's retry
    pass (16): the env is requeued and the retry pass solves it in its LDS workspace for up to 25 supporters -- LAPACK's own limit
    for this path of dgelsd (SMLSIZ).  No flag, every observation the reference's.  N_pad: the same four envs among 60 ordinary ones
    (the retry queue then holds them next to nothing / next to each other in other blocks)."""
    c, z = load_case("plate_discrete_s1")
    items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
    N = c["N"] + N_pad
    stream = np.zeros((N,) + z["stream"].shape[1:], np.int32)
    stream[:c["N"]] = z["stream"]
    stream[c["N"]:] = make_stream(5, N_pad, z["stream"].shape[1], items) if N_pad else 0
    env = _pkg().PctVecEnv(N, setting=c["setting"], container_size=c["container"], item_set=items, internal_node_holder=c["I"],
                           leaf_node_holder=c["L"], env_id_base=c["base"], item_stream=stream, device="cuda:0")
    obs = env.reset()
    retried = 0
    for t in range(c["steps"]):
        assert np.array_equal(obs.cpu().numpy()[:c["N"]], z["obs"][t]), t
        rows = np.zeros((N, 3), np.float32)
        rows[:c["N"]] = z["actions"][t]
        rows[c["N"]:, 1:] = (t % 7, (3 * t) % 7)   # the padding envs: unit-ish placements wherever (they may end episodes)
        obs, reward, done, _ = env.step(rows)
        retried += env.debug_retry_count()
        assert np.array_equal(done.astype(np.uint8)[:c["N"]], z["done"][t]), t
        assert np.array_equal(reward[:c["N"], 0].numpy(), z["reward"][t].astype(np.float32)), t
    assert np.array_equal(obs.cpu().numpy()[:c["N"]], z["obs"][c["steps"]])
    assert not (env.error_flags & 0x17).any(), env.error_flags[:c["N"]]  # no capacity flag anywhere (the notices 0x40 / 0x80 may be up)
    assert not (env.error_flags[:c["N"]] & 0x3F).any()
    assert retried > 0  # the wide splits went through the retry pass
    env.close()


@pytest.mark.parametrize("name", ["discrete_s1_ondomain_avx2", "discrete_s1_flat_lstsq_avx2"])
def test_hip_gelsd_avx2_matches_reference_on_avx2_kernels(name):
    """PCT_LSTSQ_GELSD_AVX2: the reference as it runs on AVX2 hosts, AMD Zen included (tests/golden/gen_golden_avx2.py: NumPy's OpenBLAS
    forced onto its Haswell kernel set -- np.dot unfused, dgemv 'N' / daxpy / dgemm summed differently).  The kernels in that mode
    follow the recording; in PCT_LSTSQ_GELSD mode they leave discrete_s1_ondomain_avx2 at step 115 of env 1, where the reference on an
    AVX-512 host and the reference on an AVX2 host part ways."""
    c, z = load_case(name)
    for mode in ("gelsd_avx2", "gelsd"):
        env = _pkg().PctVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                               internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"], item_stream=z["stream"],
                               device="cuda:0", lstsq=mode)
        obs = env.reset()
        first = np.full(c["N"], -1)
        for t in range(c["steps"] + 1):
            bad = (obs.cpu().numpy() != z["obs"][t]).any(1)
            first = np.where((first < 0) & bad, t, first)
            if t < c["steps"]:
                env.step_hash_policy(1)
                obs, _, _, _ = env.step_wait()
        assert not env.error_flags.any()
        env.close()
        if mode == "gelsd_avx2":
            assert (first < 0).all(), first
        else:
            assert np.array_equal(first, z["first_difference_from_avx512"]), first


@pytest.mark.parametrize("name", ["continuous_s1_flat_lstsq", "continuous_s1_unit_80_50"])
def test_hip_gelsd_continuous_matches_reference_fixture(name):
    c, z = load_case(name)
    env = _pkg().PctVecEnv(c["N"], setting=c["setting"], container_size=c["container"], continuous=True, sample_left_bound=c["lo"],
                           sample_right_bound=c["hi"], internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"],
                           item_stream=z["stream"], device="cuda:0", lstsq="gelsd")
    obs = env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(obs.cpu().numpy(), z["obs"][t].astype(np.float32)), (name, t)
        env.step_hash_policy(1)
        obs, _, _, _ = env.step_wait()
    assert not env.error_flags.any()
    env.close()


@pytest.mark.parametrize("kind", ["c1", "wide_flat", "continuous"])
def test_hip_gelsd_matches_oracle_gelsd(kind, oracle_gelsd):
    """kernels and oracle both in gelsd mode on seeded streams: observations after every step, rewards, dones, counters, ratios, and the
    commit-solve part of the notice env by env.  c1: the C1 domain at 1024 envs; wide_flat: flat items on a 20^3 bin (splits over up to 16
    supporters: the retry pass's workspace class); continuous: the unit-bin setting-1 domain."""
    if kind == "c1":
        N, steps = 1024, 150
        items = [(a, b, c) for a in range(1, 6) for b in range(1, 6) for c in range(1, 6)]
        kw = dict(setting=1, container_size=(10, 10, 10), item_set=items, internal_node_holder=80, leaf_node_holder=50, env_id_base=3)
        stream = make_stream(77, N, 1024, items)
        env = _pkg().PctVecEnv(N, item_stream=stream, device="cuda:0", lstsq="gelsd", **kw)
        ora = OracleVecEnv(N, **kw)
        ora.set_item_stream(stream)
    elif kind == "wide_flat":
        N, steps = 24, 200  # (scripts/gelsd_gpu_check.py runs 48 x 300: ~100 s, a 121 x 16 system solved by one lane is slow)
        items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
        kw = dict(setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50, env_id_base=5)
        stream = make_stream(4242, N, 2048, items)
        env = _pkg().PctVecEnv(N, item_stream=stream, device="cuda:0", lstsq="gelsd", **kw)
        ora = OracleVecEnv(N, **kw)
        ora.set_item_stream(stream)
    else:
        N, steps = 256, 120
        env = _pkg().PctVecEnv(N, setting=1, container_size=(1, 1, 1), continuous=True, sample_left_bound=0.1, sample_right_bound=0.5,
                               internal_node_holder=80, leaf_node_holder=50, seed=21, env_id_base=9, device="cuda:0", lstsq="gelsd")
        ora = OracleVecEnv(N, setting=1, container_size=(1, 1, 1), env_kind=1, sample_bounds=(0.1, 0.5), internal_node_holder=80,
                           leaf_node_holder=50, env_id_base=9)
        ora.set_sampler(21)
    obs = env.reset()
    ora.reset()
    for t in range(steps):
        assert np.array_equal(obs.cpu().numpy(), ora.obs.astype(np.float32)), (kind, t)
        env.step_hash_policy(1)
        ora.step_hash_policy(1)
        obs, rew, done, _ = env.step_wait()
        assert np.array_equal(done.astype(np.uint8), ora.done), (kind, t)
        assert np.array_equal(env._h_counter.numpy(), ora.counter), (kind, t)
        assert np.array_equal(rew[:, 0].numpy(), ora.reward.astype(np.float32)), (kind, t)
        assert np.array_equal(env._h_ratio.numpy(), ora.ratio), (kind, t)
    # The notice, by provenance.  Solves of a COMMIT walk are made by both sides whatever the evaluation order:
    # the set of envs whose commit solves raised the notice (PCT_FLAG_ILL_COMMIT / the oracle's ill_commit) must be EQUAL.  Solves of
    # VIRTUAL checks are not comparable one by one -- the reference's recursion returns at a candidate's first unstable supporter
    # (solves further on are never made), the wave examines a candidate's walk tasks side by side and drops the rest once one fails
    # (solves the recursion made first may never be made) -- so the virtual-only part is logged, and bounded: every flagged env of
    # one side that the other side does not flag at all must be a virtual-only notice there.
    gpu_ill, ora_ill = np.asarray(env.ill_conditioned, bool), ora.ill_conditioned().astype(bool)
    gpu_com, ora_com = np.asarray(env.ill_commit, bool), ora.ill_commit().astype(bool)
    print("notice: kernels %d envs (%d by a commit solve), oracle %d envs (%d by a commit solve), both %d; virtual-only difference: kernels-only %s, oracle-only %s"
          % (gpu_ill.sum(), gpu_com.sum(), ora_ill.sum(), ora_com.sum(), (gpu_ill & ora_ill).sum(),
             np.nonzero(gpu_ill & ~ora_ill)[0].tolist(), np.nonzero(ora_ill & ~gpu_ill)[0].tolist()))
    assert np.array_equal(gpu_com, ora_com), (kind, np.nonzero(gpu_com != ora_com)[0].tolist())
    assert not (gpu_com & ~gpu_ill).any() and not (ora_com & ~ora_ill).any()
    assert not ((gpu_ill & ~ora_ill) & gpu_com).any() and not ((ora_ill & ~gpu_ill) & ora_com).any()
    assert not env.error_flags.any()
    env.close()
    ora.close()


def test_hip_lstsq_mode_switches_between_steps(oracle_gelsd):
    """pct_set_lstsq_mode is callable between steps: a handle created in the default mode and switched to gelsd before the
    first reset behaves like one created with lstsq='gelsd'; setting 2 accepts the call and ignores it"""
    items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
    kw = dict(setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50)
    stream = make_stream(11, 8, 1024, items)
    pkg = _pkg()
    env = pkg.PctVecEnv(8, item_stream=stream, device="cuda:0", **kw)
    pkg._lib.check(env._L.pct_set_lstsq_mode(env._h, pkg._lib.LSTSQ_GELSD))
    ora = OracleVecEnv(8, **kw)
    ora.set_item_stream(stream)
    obs = env.reset()
    ora.reset()
    for t in range(200):
        assert np.array_equal(obs.cpu().numpy(), ora.obs.astype(np.float32)), t
        env.step_hash_policy(1)
        ora.step_hash_policy(1)
        obs, _, _, _ = env.step_wait()
    env.close()
    ora.close()
    e2 = pkg.PctVecEnv(4, setting=2, container_size=(10, 10, 10), item_set=[(1, 1, 1), (2, 2, 2)], device="cuda:0", lstsq="gelsd")
    e2.reset()
    e2.close()
