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


@pytest.mark.parametrize("mode", ["default", "jacobi"])
def test_product_stability_equals_oracle_on_random_streams(mode):
    """the product's stability source against the oracle's, both in the default solver mode (dgelsd) and both in the Jacobi stand-in
    (rounds 1-4's default, now opt-in: pct_stab.cuh's stab_split_fixed / stab_lstsq against the oracle's lstsq_min_norm)"""
    if mode == "jacobi":
        with _Gelsd(ol.LSTSQ_JACOBI):
            _product_equals_oracle(ol.LSTSQ_JACOBI)
    else:
        _product_equals_oracle(None)


def _product_equals_oracle(force_mode):
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
        old = ol.set_lstsq_mode(force_mode) if force_mode is not None else None
        try:
            b = ol.OracleVecEnv(48, setting=1, container_size=(10, 10, 10), item_set=items)
            b.set_item_stream(stream)
            b.reset()
            for t in range(600):
                assert np.array_equal(b.obs, ref_obs[t]), t
                b.step_hash_policy(1)
                assert np.array_equal(b.done, ref_done[t]), t
            b.close()
        finally:
            if old is not None:
                ol.set_lstsq_mode(old)


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


# ---- PCT_LSTSQ_GELSD: the product's dgelsd (csrc/pct_gelsd.cuh), compiled for the host ---------------------------------------
class _Gelsd(object):
    """the loaded oracle library (plain or variant) in gelsd mode"""

    def __init__(self, mode=None):
        self.mode = ol.LSTSQ_GELSD if mode is None else mode

    def __enter__(self):
        self.old = ol.set_lstsq_mode(self.mode)

    def __exit__(self, *a):
        ol.set_lstsq_mode(self.old)


@pytest.mark.parametrize("name", ["discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq", "discrete_s1_flat20", "discrete_s1_10_80_50",
                                  "discrete_s1_flat_diverging"])
def test_product_gelsd_matches_reference_fixture(name):
    """the kernels' strict solver against the unmodified reference: thousands of least-squares splits over up to eight
    supporters, and -- discrete_s1_flat_diverging -- the stream on which the default Jacobi solver parts ways at step 79"""
    c, z = load_case(name)
    with _Variant(), _Gelsd():
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                              internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        if case_density(z) is not None:
            env.set_density_stream(case_density(z))
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), (name, t)
            env.step_hash_policy(1)
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
        env.close()


@pytest.mark.parametrize("name", ["continuous_s1_flat_lstsq", "continuous_s1_unit_80_50"])
def test_product_gelsd_continuous_matches_reference_fixture(name):
    c, z = load_case(name)
    with _Variant(), _Gelsd():
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], env_kind=1,
                              sample_bounds=(c["lo"], c["hi"]), internal_node_holder=c["I"], leaf_node_holder=c["L"],
                              env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs, z["obs"][t]), (name, t)
            env.step_hash_policy(1)
        env.close()


def test_product_gelsd_equals_oracle_gelsd_on_flat_streams():
    """product source vs the oracle's own restatement (two independently written dgelsd, one with long double, one with integer
    x87 arithmetic), both in gelsd mode: observations, dones and the ill-conditioning notice over wide flat items on a 20^3 bin"""
    items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
    stream = make_stream(4242, 16, 2048, items)
    kw = dict(setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50)
    with _Gelsd():
        a = ol.OracleVecEnv(16, **kw)
        a.set_item_stream(stream)
        a.reset()
        ref_obs, ref_done, ref_ill = [], [], []
        for t in range(400):
            ref_obs.append(a.obs.copy())
            a.step_hash_policy(1)
            ref_done.append(a.done.copy())
            ref_ill.append(a.ill_conditioned().copy())
        a.close()
    with _Variant(), _Gelsd():
        b = ol.OracleVecEnv(16, **kw)
        b.set_item_stream(stream)
        b.reset()
        for t in range(400):
            assert np.array_equal(b.obs, ref_obs[t]), t
            b.step_hash_policy(1)
            assert np.array_equal(b.done, ref_done[t]), t
            assert np.array_equal(b.ill_conditioned(), ref_ill[t]), t
        b.close()


def test_product_x87_dnrm2_equals_long_double():
    """pct_gelsd.cuh emulates OpenBLAS' x87 dnrm2 (80-bit squares and sums, fsqrt, one rounding to double) with 64-bit integer
    mantissas; the oracle computes the same with the FPU's long double"""
    plain = ctypes.CDLL(ol.build())
    plain.gelsd_dnrm2.restype = ctypes.c_double
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    var.gelsd_host_dnrm2.restype = ctypes.c_double
    rng = np.random.default_rng(8)
    for t in range(20000):
        n = int(rng.integers(1, 130))
        inc = int(rng.integers(1, 4))
        x = rng.standard_normal(n * inc) * 10.0 ** (rng.integers(-140, 140) if t % 4 == 0 else rng.integers(-3, 4))
        if t % 7 == 0:
            x[rng.integers(0, len(x))] = 0.0
        if t % 5 == 0:
            x = np.round(x * 8) / 8  # many exactly representable squares / ties
        p = x.ctypes.data_as(ctypes.c_void_p)
        assert plain.gelsd_dnrm2(ctypes.c_int(n), p, ctypes.c_int(inc)) == var.gelsd_host_dnrm2(ctypes.c_int(n), p, ctypes.c_int(inc)), t


def test_product_stability_matches_reference_plate_fixture():
    """splits over 20 / 22 / 24 supporters (round 6: the cap is LAPACK's own SMLSIZ = 25, no longer 16) through the PRODUCT's
    stability source compiled for the host, against the unmodified reference's recording (tests/golden/gen_plate_golden.py)"""
    c, z = load_case("plate_discrete_s1")
    items = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
    with _Variant():
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=items, internal_node_holder=c["I"],
                              leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), t
            env.step_rows(z["actions"][t].astype(np.float64))
            assert np.array_equal(env.done, z["done"][t]) and np.array_equal(env.reward, z["reward"][t]), t
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
        assert not env.flags.any()
        env.close()


def test_product_certified_dnrm2_equals_x87_emulation():
    """round 6: dnrm2 of up to eight elements by a double-double sum of squares + one corrected square root, taken only where a
    distance-to-the-rounding-boundary certificate proves it equal to the x87 chain (pct_gelsd.cuh dnrm2_certified); everything else
    still runs the integer emulation.  2 * 10^7 vectors of the value classes the stability systems hold."""
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    out = (ctypes.c_long * 3)()
    var.gelsd_host_dnrm2_certified_sweep(ctypes.c_long(20_000_000), ctypes.c_ulonglong(2026), out)
    vectors, certified, diff = out[0], out[1], out[2]
    assert vectors == 20_000_000 and diff == 0, (vectors, certified, diff)
    assert 0.80 * vectors < certified < 0.99 * vectors, certified  # (one class of seven is beyond the certificate's exponent range)


def test_product_pow_is_glibc_pow():
    """`tri_base_len ** 2` (C/space.py:108,206; D/space.py:112,210) is libm pow(len, 2.0), which is NOT len * len (glibc's pow is
    not correctly rounded): csrc/pct_pow.cuh restates glibc's pow as its FMA build executes it.  Against the live libm: 1.2 * 10^7
    lengths of the continuous env's lever rule (unit bin, and the 100-unit bin of BASELINE configs[4]) -- 0 differences, while
    len * len differs on ~0.08 % of them; every length the discrete env can produce in bins up to 43 per axis (both np.dot
    flavours): pow(len, 2.0) == len * len there, which is what the 5-bit-coordinate discrete kernels compute (bins up to 31), while
    the 10-bit-coordinate kernels run the restatement (checked up to 1023 per axis); and 10^5 general (x, y) pairs."""
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    out = (ctypes.c_long * 3)()
    for span, count in ((1.0, 8_000_000), (100.0, 4_000_000)):
        var.pow_host_sweep_continuous(ctypes.c_long(count), ctypes.c_ulonglong(77 + int(span)), ctypes.c_double(span), out)
        assert out[0] > 0.99 * count and out[1] == 0, (span, out[0], out[1])
        assert out[2] > 0, "the sweep no longer sees pow(len, 2) != len * len: is libm's pow folded away?"
    # the 5-bit-coordinate discrete kernels (bins up to 31 per axis) compute len * len: equal to pow on every length up to 43 per axis
    var.pow_host_sweep_discrete(ctypes.c_int(43), out)
    assert out[0] == 2 * (87 * 87 - 1) and out[1] == 0 and out[2] == 0, (out[0], out[1], out[2])
    # larger bins (the 10-bit-coordinate kernels) run the restatement: pow != len * len from (39.5, 43.5) on
    var.pow_host_sweep_discrete(ctypes.c_int(1023), out)
    assert out[1] == 0 and out[2] > 0, (out[0], out[1], out[2])
    var.pow_host.restype = ctypes.c_double
    var.pow_host.argtypes = [ctypes.c_double, ctypes.c_double]
    rng = np.random.default_rng(5)
    xs = np.exp(rng.uniform(-8, 8, 100_000))
    ys = rng.uniform(-4, 4, 100_000)
    ys[::5] = 2.0
    import math
    for x, y in zip(xs.tolist(), ys.tolist()):
        assert var.pow_host(x, y) == math.pow(x, y), (x, y)


def test_product_gelsd_split_equals_recorded_numpy_solutions():
    """the product's split (geometry -> system -> dgelsd) on geometry whose dot products are exact in double on any machine
    (half-integer coordinates), against the oracle's restatement fed the same system -- which tests/test_gelsd_port.py pins to NumPy"""
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    rng = np.random.default_rng(12)
    deficient = 0
    for t in range(3000):
        k = int(rng.choice([3, 3, 4, 5, 6, 8, 11, 16]))
        pts = rng.integers(0, 40 if t % 2 else 6, (k, 2)) / 2.0  # (the narrow grid: collinear / repeated centres, dropped rows)
        com = rng.integers(0, 80 if t % 2 else 12, 2) / 4.0
        M = k * (k - 1) // 2 + 1
        A = np.zeros((M, k))
        b = np.zeros(M)
        r = 0
        for i in range(k - 1):
            for j in range(i + 1, k):
                tv = pts[i] - pts[j]
                mol = float((com - pts[i]) @ tv)
                if mol != 0:
                    A[r, i] = 1
                    A[r, j] = -abs(float((com - pts[j]) @ tv)) / mol
                r += 1
        A[-1, :] = 1
        b[-1] = 1
        x, rank, sv, near = ol.gelsd_lstsq(A, b)
        deficient += rank < k
        x2 = np.zeros(k)
        ill = ctypes.c_int(0)
        cen = np.ascontiguousarray(pts)
        ok = var.gelsd_host_split(ctypes.c_int(k), cen.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(com[0]), ctypes.c_double(com[1]),
                                  x2.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ill))
        assert ok and np.array_equal(x, x2) and bool(ill.value) == near, (t, k)
    assert deficient > 5


def test_product_dbdsqr_equals_oracle_dbdsqr():
    """the bidiagonal SVD sweeps (zero-shift and shifted, chasing downwards and upwards -- written once for both directions in the
    product, four times as in LAPACK in the oracle): singular values, V^T and the rotated right-hand side, bit for bit, on graded,
    split and random bidiagonals"""
    plain = ctypes.CDLL(ol.build())
    plain.gelsd_dbdsqr.restype = ctypes.c_int
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    rng = np.random.default_rng(31)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for t in range(6000):
        n = int(rng.integers(1, 17))
        d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 1))
        if t % 3 == 0:
            d = d * 10.0 ** rng.integers(-9, 1, n)  # graded: zero-shift sweeps
        if t % 4 == 0:
            d = d[::-1].copy()                       # larger end at the bottom: the upward chase
        if t % 5 == 0:
            e[rng.integers(0, len(e))] = 0.0         # a split
        if t % 7 == 0:
            d[rng.integers(0, n)] = 0.0
        vt, c = np.asfortranarray(np.eye(n)), rng.standard_normal(n)
        d2, e2, vt2, c2 = d.copy(), e.copy(), vt.copy(order="F"), c.copy()
        i1 = plain.gelsd_dbdsqr(ctypes.c_int(n), ctypes.c_int(n), P(d), P(e), P(vt), ctypes.c_int(n), P(c), P(np.zeros(4 * n + 8)))
        i2 = var.gelsd_host_dbdsqr(ctypes.c_int(n), P(d2), P(e2), P(vt2), P(c2), P(np.zeros(4 * n + 8)))
        assert i1 == i2 and np.array_equal(d, d2) and np.array_equal(vt, vt2) and np.array_equal(c, c2), (t, n)


def test_product_dbdsqr3_equals_generic_dbdsqr():
    """round 6: dbdsqr for n = 3 with d, e and a sweep's rotations in registers and every index static (pct_gelsd.cuh dbdsqr3 -- what
    the kernels run for three supporters, 94 % of the solves) against the generic LDS-resident routine, 2 * 10^6 random bidiagonals
    (graded, reversed, split, nearly diagonal, with zero diagonal entries, with ties): every bit of d, VT, the rotated column and the
    return value"""
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    out = (ctypes.c_long * 2)()
    var.gelsd_host_dbdsqr3_sweep(ctypes.c_long(2_000_000), ctypes.c_ulonglong(31), out)
    assert out[0] == 2_000_000 and out[1] == 0, (out[0], out[1])


@pytest.mark.parametrize("name", ["discrete_s1_ondomain_avx2", "discrete_s1_flat_lstsq_avx2"])
def test_product_gelsd_avx2_matches_reference_on_avx2_kernels(name):
    """PCT_LSTSQ_GELSD_AVX2 in the product's source: the reference as it runs on AVX2 hosts (tests/golden/gen_golden_avx2.py)"""
    c, z = load_case(name)
    with _Variant(), _Gelsd(ol.LSTSQ_GELSD_AVX2):
        env = ol.OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=case_items(c),
                              internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
        env.set_item_stream(z["stream"])
        env.reset()
        for t in range(c["steps"]):
            assert np.array_equal(env.obs.astype(np.float32), z["obs"][t]), (name, t)
            env.step_hash_policy(1)
        assert np.array_equal(env.obs.astype(np.float32), z["obs"][c["steps"]])
        env.close()


def test_product_gelsd_avx2_split_equals_oracle_avx2():
    """both flavours of the product's split against the oracle's restatement with the same kernel set, on geometry with exact dot
    products; and the two flavours differ from each other on most systems"""
    with _Variant():
        var = ctypes.CDLL(VARIANT)
    plain = ctypes.CDLL(ol.build())
    rng = np.random.default_rng(13)
    differ = total = 0
    for t in range(1500):
        k = int(rng.choice([3, 4, 5, 6, 8, 9, 12, 16]))
        pts = rng.integers(0, 40, (k, 2)) / 2.0
        com = rng.integers(0, 80, 2) / 4.0
        M = k * (k - 1) // 2 + 1
        A = np.zeros((M, k))
        b = np.zeros(M)
        r = 0
        for i in range(k - 1):
            for j in range(i + 1, k):
                tv = pts[i] - pts[j]
                mol = float((com - pts[i]) @ tv)
                if mol != 0:
                    A[r, i] = 1
                    A[r, j] = -abs(float((com - pts[j]) @ tv)) / mol
                r += 1
        A[-1, :] = 1
        b[-1] = 1
        xs = []
        for mode in (1, 2):
            plain.gelsd_set_kernel_set(mode - 1)
            x, rank, sv, near = ol.gelsd_lstsq(A, b)
            var.stab_set_lstsq_mode(mode)
            x2 = np.zeros(k)
            ill = ctypes.c_int(0)
            cen = np.ascontiguousarray(pts)
            ok = var.gelsd_host_split(ctypes.c_int(k), cen.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(com[0]), ctypes.c_double(com[1]),
                                      x2.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ill))
            assert ok and np.array_equal(x, x2), (t, k, mode)
            xs.append(x2)
        total += 1
        differ += not np.array_equal(xs[0], xs[1])
    plain.gelsd_set_kernel_set(0)
    var.stab_set_lstsq_mode(1)  # (the default)
    assert differ > 0.5 * total
