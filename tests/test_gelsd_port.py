"""oracle/pct_oracle_gelsd.c -- np.linalg.lstsq as the reference's NumPy executes it (LAPACK dgelsd, operation for operation).

The reference solves the >= 3-supporter split of the stability check with np.linalg.lstsq (D/space.py:152,249; C/space.py:148,245).
tests/golden/check_gelsd_port.py pins the restatement to the live library routine by routine in the build container; here, on any
machine: the committed vectors (tests/golden/lstsq_systems.npz: systems recorded from reference runs and constructed ones, with
NumPy's solution / rank / singular values), and the reference fixtures through the oracle in gelsd mode -- including the adversarial
stream on which the Jacobi stand-in parts ways with the reference (discrete_s1_flat_diverging)."""
import os

import numpy as np
import pytest

from oracle import oracle_lib
from oracle.oracle_lib import OracleVecEnv
from tests.common import case_items, load_case

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture
def gelsd_mode():
    old = oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_GELSD)
    yield
    oracle_lib.set_lstsq_mode(old)


FLAVOURS = [("lstsq_systems.npz", oracle_lib.LSTSQ_GELSD), ("lstsq_systems_avx2.npz", oracle_lib.LSTSQ_GELSD_AVX2)]


def _systems(fname="lstsq_systems.npz"):
    z = np.load(os.path.join(HERE, "golden", fname))
    off = 0
    for i in range(len(z["M"])):
        m, n = int(z["M"][i]), int(z["N"][i])
        a = z["A"][off:off + m * n].reshape(m, n)
        off += m * n
        b = np.zeros(m)
        b[-1] = 1.0
        yield a, b, z["x"][i, :n], int(z["rank"][i]), z["sv"][i, :n]


@pytest.mark.parametrize("fname,mode", FLAVOURS)
def test_gelsd_port_reproduces_numpy_bit_for_bit(fname, mode):
    """x, the effective rank and the singular values of every committed system: identical to what NumPy 2.2.6 (OpenBLAS 0.3.29)
    returned in the build container -- 3 .. 16 supporters, rank-deficient systems included -- with its AVX-512 kernel set
    (lstsq_systems.npz, LSTSQ_GELSD) and with the kernel set of AVX2 hosts (OPENBLAS_CORETYPE=HASWELL: lstsq_systems_avx2.npz,
    LSTSQ_GELSD_AVX2).  The two recordings are the same systems; their solutions differ in the last bits on most of them."""
    old = oracle_lib.set_lstsq_mode(mode)  # (selects the kernel set the direct call below uses)
    oracle_lib.set_lstsq_mode(old)
    import ctypes
    ctypes.CDLL(oracle_lib.build()).gelsd_set_kernel_set(1 if mode == oracle_lib.LSTSQ_GELSD_AVX2 else 0)
    n = deficient = 0
    sizes = set()
    try:
        for a, b, x, rank, sv in _systems(fname):
            x2, rank2, sv2, _ = oracle_lib.gelsd_lstsq(a, b)
            assert np.array_equal(x, x2), (n, a.shape)
            assert rank == rank2 and np.array_equal(sv, sv2), (n, a.shape)
            n += 1
            deficient += rank < a.shape[1]
            sizes.add(a.shape[1])
    finally:
        ctypes.CDLL(oracle_lib.build()).gelsd_set_kernel_set(0)
    assert n > 1000 and deficient > 0 and {3, 4, 5, 6, 8, 16} <= sizes


def test_the_two_kernel_sets_give_numpy_other_last_bits():
    """the reference's np.linalg.lstsq is a property of the machine: of the same 1066 systems the AVX-512 and the AVX2 recording
    agree bit for bit on a minority"""
    same = total = 0
    for (a, _, x, _, _), (a2, _, x2, _, _) in zip(_systems("lstsq_systems.npz"), _systems("lstsq_systems_avx2.npz")):
        if np.array_equal(a, a2):
            total += 1
            same += np.array_equal(x, x2)
    assert total > 500 and same < 0.5 * total, (same, total)


def test_gelsd_port_is_a_least_squares_solution():
    """independent of the recorded vectors: the normal equations hold and the solution is the minimum-norm one (checked
    against NumPy's solver of whatever machine this runs on, to 1e-9 -- the last bits are the machine's)"""
    for i, (a, b, _, rank, _) in enumerate(_systems()):
        if i % 7:
            continue
        x2, rank2, sv2, _ = oracle_lib.gelsd_lstsq(a, b)
        want = np.linalg.lstsq(a, b, rcond=None)
        if int(want[2]) == rank:  # (a rank decision at the cut may fall differently on another machine)
            assert np.allclose(x2, want[0], rtol=1e-7, atol=1e-9 * max(1.0, np.abs(want[0]).max())), i


@pytest.mark.parametrize("name", ["discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq", "discrete_s1_flat20"])
def test_oracle_gelsd_mode_matches_reference_fixtures(name, gelsd_mode):
    """the adversarial flat-item fixtures (thousands of least-squares splits, up to eight supporters): unmodified reference ==
    oracle in gelsd mode, every observation / reward / done"""
    c, z = load_case(name)
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    if "density" in z:
        env.set_density_stream(z["density"])
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), t
        env.step_hash_policy(1)
        assert np.array_equal(env.reward, z["reward"][t]) and np.array_equal(env.done, z["done"][t]), t
    env.close()


def test_oracle_gelsd_mode_follows_the_reference_through_the_rank_cut(gelsd_mode):
    """discrete_s1_flat_diverging: the unmodified reference on the stream whose env 0 meets a rank decision 1.97 x above the rcond
    cut at step 79 -- the Jacobi stand-in parts ways there (test_oracle_notice_precedes_the_lapack_divergence); with dgelsd
    restated the oracle IS the reference on all four envs over the whole recording"""
    c, z = load_case("discrete_s1_flat_diverging")
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(z["stream"])
    env.reset()
    for t in range(c["steps"]):
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), t
        env.step_hash_policy(1)
    assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
    assert env.ill_conditioned()[0]  # the notice is still raised: a singular value within 1e3 of the cut
    env.close()


@pytest.mark.parametrize("name", ["discrete_s1_ondomain_avx2", "discrete_s1_flat_lstsq_avx2"])
def test_oracle_avx2_flavour_matches_reference_on_avx2_kernels(name):
    """tests/golden/gen_golden_avx2.py: the unmodified reference with NumPy's OpenBLAS forced onto the kernel set of AVX2 hosts (AMD Zen
    included).  LSTSQ_GELSD_AVX2 follows it; on discrete_s1_ondomain_avx2 the AVX-512 flavour leaves that recording at step 115 of env
    1 -- exactly where the reference on an AVX-512 host and the reference on an AVX2 host part ways (profiles/r04_lstsq_ondomain.txt)"""
    c, z = load_case(name)
    for mode in (oracle_lib.LSTSQ_GELSD_AVX2, oracle_lib.LSTSQ_GELSD):
        old = oracle_lib.set_lstsq_mode(mode)
        try:
            env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                               internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
            env.set_item_stream(z["stream"])
            env.reset()
            first = np.full(c["N"], -1)
            for t in range(c["steps"] + 1):
                bad = (env.obs.astype(np.float32) != z["obs"][t]).any(1)
                first = np.where((first < 0) & bad, t, first)
                if t < c["steps"]:
                    env.step_hash_policy(1)
            env.close()
        finally:
            oracle_lib.set_lstsq_mode(old)
        if mode == oracle_lib.LSTSQ_GELSD_AVX2:
            assert (first < 0).all(), first
        else:
            assert np.array_equal(first, z["first_difference_from_avx512"]), first


@pytest.mark.parametrize("name,heur", [("heur_s1_10", "LSAH"), ("heur_s1_10", "OnlineBPH"), ("heur_s1_10", "DBL"), ("heur_s1_10", "BR"),
                                       ("heur_macs_s1_rect", "MACS")])
def test_oracle_gelsd_mode_heuristics_match_reference_loops(name, heur, gelsd_mode):
    """heuristic.py's baselines under the stability setting (their feasibility probes run the same check): per-episode utilisation and
    length of the unmodified reference's loops, with the splits solved as dgelsd solves them"""
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
    env.close()


def test_host_flavour_helper_names_this_numpys_kernel_set():
    """online-3d-bpp-pct_amd/lstsq_mode.py: PctVecEnv(lstsq="numpy") = the flavour of the NumPy this process runs (for comparing
    against a reference in the same Python).  The two fixture files say which kernel set recorded them; the helper must map those
    names, and on this machine must return a mode whose recorded solutions np.linalg.lstsq reproduces right now."""
    import importlib
    lm = importlib.import_module("online-3d-bpp-pct_amd.lstsq_mode")
    assert lm.RESTATED == {"SkylakeX": "gelsd", "Haswell": "gelsd_avx2"}
    arch, version = lm.numpy_blas()
    mode = lm.numpy_lstsq_mode()
    assert mode == lm.RESTATED.get(arch)
    if mode is None or version != lm.PINNED_OPENBLAS:
        pytest.skip("this host's NumPy runs %r kernels of OpenBLAS %r: nothing recorded to compare with" % (arch, version))
    fname = "lstsq_systems.npz" if mode == "gelsd" else "lstsq_systems_avx2.npz"
    n = 0
    for a, b, x, rank, sv in _systems(fname):
        if n % 5 == 0:
            want = np.linalg.lstsq(a, b, rcond=None)
            assert np.array_equal(np.asarray(want[0]).ravel(), x) and int(want[2]) == rank, n
        n += 1
