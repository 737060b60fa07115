"""CPU: the product's restructured stability check (csrc/pct_stab.cuh -- the code the GPU lanes
run), compiled for the host by tests/host/, against the reference fixtures, the setting-1
known answer and the oracle's own (recursive, dictionary-shaped) restatement."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle.oracle_lib as ol
from tests.common import case_items, case_density, CONT_STAB_CASES, GOLDEN, STAB_CASES, item_set_range, load_case, make_stream

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANT = os.path.join(HERE, "host", "libpct_oracle_prodstab.so")


class _Variant(object):
    """Temporarily points oracle_lib at the product-stability variant of the oracle."""

    def __enter__(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "-s"])
        self.saved = (ol._LIB, ol.build)
        ol._LIB = None
        ol.build = lambda force=False: VARIANT
        return self

    def __exit__(self, *a):
        ol._LIB, ol.build = self.saved


@pytest.mark.parametrize("name", STAB_CASES)
def test_product_stability_matches_reference_fixture(name):
    c, z = load_case(name)
    with _Variant():
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                              internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        if case_density(z) is not None:
            env.set_density_stream(case_density(z))
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), (name, t)
            env.step_hash_policy(1)
            assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t])
        env.close()


def test_product_stability_known_answer():
    z = np.load(GOLDEN + "/kat_discrete_s1.npz")
    with _Variant():
        env = ol.OracleVecEnv(1, setting=1, container_size=(10, 10, 10), item_set=item_set_range(1, 5))
        env.set_item_stream(z["items"][None])
        env.reset()
        h = hashlib.sha256()
        for t in range(500):
            h.update(env.obs[0].astype(np.float32).tobytes())
            env.step_rows(z["actions"][t][None].astype(np.float64))
        env.close()
    assert h.hexdigest()[:16] == "443198ae2c0162db"


def test_product_stability_equals_oracle_on_random_streams():
    items = item_set_range(1, 5)
    stream = make_stream(99, 48, 1024, items)
    a = ol.OracleVecEnv(48, setting=1, container_size=(10, 10, 10), item_set=items)
    a.set_item_stream(stream)
    a.reset()
    ref_obs, ref_done = [], []
    for t in range(600):
        ref_obs.append(a.obs.copy())
        a.step_hash_policy(1)
        ref_done.append(a.done.copy())
    a.close()
    with _Variant():
        b = ol.OracleVecEnv(48, setting=1, container_size=(10, 10, 10), item_set=items)
        b.set_item_stream(stream)
        b.reset()
        for t in range(600):
            assert np.array_equal(b.obs, ref_obs[t]), t
            b.step_hash_policy(1)
            assert np.array_equal(b.done, ref_done[t]), t
        b.close()


@pytest.mark.parametrize("name", CONT_STAB_CASES)
def test_product_stability_continuous_matches_reference_fixture(name):
    """Continuous env, setting 1: the product's stability code (1e-6 margins, rounded contact
    rectangles) inside the continuous oracle, against the float64 reference fixture."""
    c, z = load_case(name)
    with _Variant():
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], env_kind=1,
                              sample_bounds=(c["lo"], c["hi"]), internal_node_holder=c["I"], leaf_node_holder=c["L"],
                              env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        if case_density(z) is not None:
            env.set_density_stream(case_density(z))
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs, z["obs"][t]), (name, t)
            env.step_hash_policy(1)
            assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t])
        env.close()
