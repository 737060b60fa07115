"""ctypes binding of the CPU oracle (oracle/libpct_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class PctConfig(ctypes.Structure):
    """Mirror of `pct_config` in include/pct_env.h."""
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("env_kind", ctypes.c_int32),
        ("setting", ctypes.c_int32),
        ("num_envs", ctypes.c_int32),
        ("container", ctypes.c_int32 * 3),
        ("internal_node_holder", ctypes.c_int32),
        ("leaf_node_holder", ctypes.c_int32),
        ("lnes", ctypes.c_int32),
        ("env_id_base", ctypes.c_int32),
        ("ems_capacity", ctypes.c_int32),
        ("candidate_capacity", ctypes.c_int32),
        ("shuffle", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libpct_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pct_oracle.c", "pct_oracle_cont.c", "pct_oracle_stab.c", "pct_oracle_gelsd.c", "pct_oracle.h",
                                               "pct_oracle_internal.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "pct_env.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpct_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        vp = ctypes.c_void_p
        L.pcto_create.argtypes = [ctypes.POINTER(PctConfig), ctypes.POINTER(vp)]
        L.pcto_destroy.argtypes = [vp]
        L.pcto_last_error.restype = ctypes.c_char_p
        L.pcto_set_item_set.argtypes = [vp, vp, ctypes.c_int32]
        L.pcto_set_item_stream.argtypes = [vp, vp, ctypes.c_int64]
        L.pcto_set_sample_bounds.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
        L.pcto_set_item_dataset.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32]
        L.pcto_set_sampler.argtypes = [vp, ctypes.c_uint64]
        L.pcto_set_numpy_rng.argtypes = [vp, ctypes.c_uint32]
        L.pcto_set_numpy_item_count.argtypes = [vp, ctypes.c_int32]
        L.pcto_set_shuffle_seed.argtypes = [vp, ctypes.c_uint64]
        L.pcto_step_heuristic.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
        L.pcto_ill_conditioned.argtypes = [vp, vp]
        L.pcto_ill_commit.argtypes = [vp, vp]
        L.pcto_set_density_stream.argtypes = [vp, vp, ctypes.c_int64]
        L.pcto_set_dataset_density.argtypes = [vp, vp]
        for name in ("pcto_obs", "pcto_reward", "pcto_done", "pcto_info_counter", "pcto_info_ratio",
                     "pcto_error_flags"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = vp
        L.pcto_reset.argtypes = [vp, vp, ctypes.c_int32]
        L.pcto_step_rows.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32]
        L.pcto_step_index.argtypes = [vp, vp, ctypes.c_int32]
        L.pcto_step_hash_policy.argtypes = [vp, ctypes.c_int32]
        L.pcto_debug_state.argtypes = [vp, ctypes.c_int32, vp, vp, ctypes.c_int32, vp, vp, vp, vp]
        L.pcto_pyset_order.argtypes = [vp, ctypes.c_int32, vp]
        L.pcto_set_num_threads.argtypes = [ctypes.c_int]
        L.pcto_set_lstsq_mode.argtypes = [ctypes.c_int]
        L.pcto_get_lstsq_mode.restype = ctypes.c_int
        L.gelsd_lstsq.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.gelsd_lstsq.restype = ctypes.c_int
        if os.environ.get("PCT_ORACLE_LSTSQ", "") in ("jacobi", "gelsd", "gelsd_avx2"):
            L.pcto_set_lstsq_mode({"jacobi": 0, "gelsd": 1, "gelsd_avx2": 2}[os.environ["PCT_ORACLE_LSTSQ"]])
        _LIB = L
    return _LIB


LSTSQ_JACOBI, LSTSQ_GELSD, LSTSQ_GELSD_AVX2 = 0, 1, 2


def set_lstsq_mode(mode):
    """Process-wide solver behind np.linalg.lstsq in the oracle's stability check: LSTSQ_GELSD (default, as the kernels: LAPACK dgelsd as
    the reference's NumPy executes it on AVX-512 hosts, oracle/pct_oracle_gelsd.c), LSTSQ_JACOBI (the stand-in of rounds 1-4) or LSTSQ_GELSD_AVX2
    (the same with the kernel set OpenBLAS runs on AVX2 hosts, AMD Zen included).  Returns the old mode."""
    old = lib().pcto_get_lstsq_mode()
    lib().pcto_set_lstsq_mode(int(mode))
    return old


def gelsd_lstsq(a, b):
    """np.linalg.lstsq(a, b, rcond=None) through oracle/pct_oracle_gelsd.c -> (x, rank, singular values, near_cut)"""
    import numpy as np
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(np.asarray(b, np.float64).ravel())
    m, n = a.shape
    x = np.zeros(n)
    sv = np.zeros(n)
    rank = ctypes.c_int(0)
    near = ctypes.c_int(0)
    info = lib().gelsd_lstsq(a.ctypes.data, b.ctypes.data, m, n, x.ctypes.data, ctypes.addressof(rank), sv.ctypes.data,
                             ctypes.addressof(near))
    if info:
        raise RuntimeError("dbdsqr did not converge")
    return x, rank.value, sv, bool(near.value)


def _np_view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (ctypes.c_byte * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleVecEnv(object):
    """Batched CPU oracle with the same call surface as the HIP handle."""

    def __init__(self, num_envs, setting=2, container_size=(10, 10, 10), item_set=None,
                 internal_node_holder=80, leaf_node_holder=50, env_kind=0, lnes=0, env_id_base=0,
                 threads=1, sample_bounds=None, shuffle=False, shuffle_seed=0):
        """env_kind 0: discrete (container / item_set in integer units).
        env_kind 1: continuous -- container in bin units (integers), sample_bounds=(left,right)
        in bin units (lattice 1e-3); item streams are int lattice units (1e-3)."""
        L = lib()
        cfg = PctConfig()
        cfg.struct_size = ctypes.sizeof(PctConfig)
        cfg.env_kind = env_kind
        cfg.setting = setting
        cfg.num_envs = num_envs
        scale = 1000 if env_kind == 1 else 1
        cfg.container[:] = [int(round(c * scale)) for c in container_size]
        cfg.internal_node_holder = internal_node_holder
        cfg.leaf_node_holder = leaf_node_holder
        cfg.lnes = lnes
        cfg.env_id_base = env_id_base
        cfg.shuffle = 1 if shuffle else 0
        self.cfg = cfg
        self.N, self.I, self.L = num_envs, internal_node_holder, leaf_node_holder
        self.row_len = (self.I + self.L + 1) * 9
        self.A = max(int(container_size[0]), int(container_size[1]))
        self._h = ctypes.c_void_p()
        self._check(L.pcto_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        L.pcto_set_num_threads(threads)
        L.pcto_set_shuffle_seed(self._h, ctypes.c_uint64(shuffle_seed))
        if env_kind == 1 and sample_bounds is not None:
            lo, hi = sample_bounds
            self._check(L.pcto_set_sample_bounds(self._h, int(round(lo * 1000)), int(round(hi * 1000))))
        elif env_kind == 1:  # not sample_from_distribution: items from item_set (bin units -> 1e-3 lattice)
            items = np.ascontiguousarray(np.rint(np.asarray(item_set, dtype=np.float64).reshape(-1, 3) * 1000).astype(np.int32))
            self._check(L.pcto_set_item_set(self._h, items.ctypes.data, items.shape[0]))
        else:
            items = np.ascontiguousarray(np.asarray(item_set, dtype=np.int32).reshape(-1, 3))
            self._check(L.pcto_set_item_set(self._h, items.ctypes.data, items.shape[0]))
        self.obs = _np_view(L.pcto_obs(self._h), (self.N, self.row_len), np.float64)
        self.reward = _np_view(L.pcto_reward(self._h), (self.N,), np.float64)
        self.done = _np_view(L.pcto_done(self._h), (self.N,), np.uint8)
        self.counter = _np_view(L.pcto_info_counter(self._h), (self.N,), np.int32)
        self.ratio = _np_view(L.pcto_info_ratio(self._h), (self.N,), np.float64)
        self.flags = _np_view(L.pcto_error_flags(self._h), (self.N,), np.uint32)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("oracle error %d: %s" % (rc, lib().pcto_last_error().decode()))

    def set_item_stream(self, items):
        items = np.ascontiguousarray(np.asarray(items, dtype=np.int32))
        assert items.ndim == 3 and items.shape[0] == self.N and items.shape[2] == 3
        self._check(lib().pcto_set_item_stream(self._h, items.ctypes.data, items.shape[1]))

    def set_item_dataset(self, trajectories, densities=None):
        """trajectories: list of [len_i,3] int arrays (lattice units), LoadBoxCreator semantics;
        densities (setting 3): list of [len_i] float arrays, the dataset's fourth column."""
        items, lengths = pack_dataset(trajectories)
        self._check(lib().pcto_set_item_dataset(self._h, items.ctypes.data, lengths.ctypes.data, items.shape[0],
                                                items.shape[1]))
        if densities is not None:
            den = pack_densities(densities, items.shape[1])
            self._check(lib().pcto_set_dataset_density(self._h, den.ctypes.data))

    def set_density_stream(self, den):
        """setting 3: den [N,T] float64, density of each env's c-th observation (c % T)."""
        den = np.ascontiguousarray(np.asarray(den, dtype=np.float64))
        assert den.ndim == 2 and den.shape[0] == self.N
        self._check(lib().pcto_set_density_stream(self._h, den.ctypes.data, den.shape[1]))

    def step_heuristic(self, kind, n_steps=1):
        self._check(lib().pcto_step_heuristic(self._h, int(kind), int(n_steps)))

    def ill_conditioned(self):
        """bool [N]: the env has taken a least-squares split whose rank decision lay within a factor 1000 of the cut
        (sticky; what the product reports as PCT_FLAG_ILL_CONDITIONED)"""
        out = np.zeros(self.N, np.uint8)
        self._check(lib().pcto_ill_conditioned(self._h, out.ctypes.data))
        return out.astype(bool)

    def ill_commit(self):
        """bool [N]: ... and a solve of a COMMIT walk raised it (PCT_FLAG_ILL_COMMIT): the part of the notice that does not depend on
        the order in which a candidate's virtual check examines its supporters"""
        out = np.zeros(self.N, np.uint8)
        self._check(lib().pcto_ill_commit(self._h, out.ctypes.data))
        return out.astype(bool)

    def set_sampler(self, seed):
        self._check(lib().pcto_set_sampler(self._h, seed))

    def set_numpy_rng(self, seed, n_item_set=125):
        """strict NumPy-stream mode: env e consumes np.random.seed(seed + env_id_base + e)'s MT19937 stream
        (continuous env: n_item_set = len(item_set) behind RandomBoxCreator's unused randint draws)"""
        if self.cfg.env_kind == 1:
            self._check(lib().pcto_set_numpy_item_count(self._h, int(n_item_set)))
        self._check(lib().pcto_set_numpy_rng(self._h, int(seed) & 0xFFFFFFFF))

    def reset(self, env_ids=None):
        if env_ids is None:
            self._check(lib().pcto_reset(self._h, None, 0))
        else:
            ids = np.ascontiguousarray(np.asarray(env_ids, dtype=np.int32))
            self._check(lib().pcto_reset(self._h, ids.ctypes.data, ids.size))
        return self.obs

    def step_rows(self, rows, auto_reset=True):
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.float64))
        assert rows.shape[0] == self.N
        self._check(lib().pcto_step_rows(self._h, rows.ctypes.data, rows.shape[1], int(auto_reset)))
        return self.obs, self.reward, self.done

    def step_index(self, idx, auto_reset=True):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
        self._check(lib().pcto_step_index(self._h, idx.ctypes.data, int(auto_reset)))
        return self.obs, self.reward, self.done

    def step_hash_policy(self, n_steps=1):
        self._check(lib().pcto_step_hash_policy(self._h, n_steps))
        return self.obs, self.reward, self.done

    def debug_state(self, e, cap_ems=1024):
        hm = np.zeros(self.A * self.A, np.int32)
        ems = np.zeros((cap_ems, 6), np.int32)
        n_ems = ctypes.c_int32()
        n_boxes = ctypes.c_int32()
        nxt = np.zeros(3, np.int32)
        cur = ctypes.c_int64()
        self._check(lib().pcto_debug_state(self._h, e, hm.ctypes.data, ems.ctypes.data, cap_ems,
                                           ctypes.byref(n_ems), ctypes.byref(n_boxes), nxt.ctypes.data,
                                           ctypes.byref(cur)))
        return dict(heightmap=hm.reshape(self.A, self.A), ems=ems[:n_ems.value].copy(), n_boxes=n_boxes.value,
                    next_item=nxt, cursor=cur.value)

    def close(self):
        if self._h:
            lib().pcto_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_dataset(trajectories):
    n = len(trajectories)
    max_len = max(len(t) for t in trajectories)
    items = np.zeros((n, max_len, 3), np.int32)
    lengths = np.zeros(n, np.int32)
    for i, t in enumerate(trajectories):
        a = np.asarray(t, dtype=np.int32).reshape(-1, 3)
        items[i, :len(a)] = a
        lengths[i] = len(a)
    return np.ascontiguousarray(items), np.ascontiguousarray(lengths)


def pack_densities(densities, max_len):
    den = np.ones((len(densities), max_len), np.float64)
    for i, d in enumerate(densities):
        d = np.asarray(d, dtype=np.float64).reshape(-1)
        den[i, :len(d)] = d
    return np.ascontiguousarray(den)


def pyset_order(keys):
    keys = np.ascontiguousarray(np.asarray(keys, dtype=np.int64).reshape(-1, 6))
    out = np.zeros(keys.shape[0] + 1, np.int32)
    n = lib().pcto_pyset_order(keys.ctypes.data, keys.shape[0], out.ctypes.data)
    return out[:n].copy()
