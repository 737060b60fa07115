#!/usr/bin/env python
"""Torch-free GPU check of PCT_LSTSQ_GELSD (seconds to start: no `import torch`, the C ABI through ctypes and libamdhip64 for the
copies).  The same comparisons as tests/test_zz_gpu_gelsd.py, for a GPU call with little time to spare:

  1. discrete_s1_flat_diverging (the unmodified reference's recording): the kernels in gelsd mode follow it to the end
     (and discrete_s1_ondomain_avx2, the reference on AVX2-host kernels, in PCT_LSTSQ_GELSD_AVX2 mode);
  2. kernels vs the oracle, both in gelsd mode: the C1 domain, wide flat items on a 20^3 bin (splits over up to 16 supporters),
     the continuous unit bin -- observation after every step, done, the notice.

    python scripts/gelsd_gpu_check.py            (exit code 0: every line PASS)
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib  # noqa: E402
from oracle.oracle_lib import OracleVecEnv  # noqa: E402
from tests.common import case_items, load_case, make_stream  # noqa: E402

vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


class Cfg(ctypes.Structure):  # include/pct_env.h pct_config
    _fields_ = [("struct_size", i32), ("env_kind", i32), ("setting", i32), ("num_envs", i32), ("container", i32 * 3),
                ("internal_node_holder", i32), ("leaf_node_holder", i32), ("lnes", i32), ("env_id_base", i32),
                ("ems_capacity", i32), ("candidate_capacity", i32), ("shuffle", i32), ("reserved", i32 * 3)]


HIP = ctypes.CDLL("libamdhip64.so")
L = ctypes.CDLL(os.path.join(ROOT, "online-3d-bpp-pct_amd", "libpct_hip.so"))
for nm in ("pct_obs", "pct_done", "pct_error_flags", "pct_info_counter"):
    getattr(L, nm).argtypes = [vp]
    getattr(L, nm).restype = vp
L.pct_last_error.restype = ctypes.c_char_p
L.pct_reset.argtypes = [vp, vp, i32, vp]
L.pct_step_hash_policy.argtypes = [vp, i32, vp]
L.pct_set_item_set.argtypes = [vp, vp, i32]
L.pct_set_item_stream.argtypes = [vp, vp, i64]
L.pct_set_sample_bounds.argtypes = [vp, i32, i32]
L.pct_set_sampler.argtypes = [vp, ctypes.c_uint64]
L.pct_set_lstsq_mode.argtypes = [vp, i32]
HIP.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]


def ck(rc):
    if rc:
        raise RuntimeError("pct error %d: %s" % (rc, L.pct_last_error().decode()))


class Env(object):
    def __init__(self, N, setting, container, I, Lh, base, continuous=False, mode=1):
        c = Cfg()
        c.struct_size = ctypes.sizeof(Cfg)
        c.env_kind = 1 if continuous else 0
        c.setting, c.num_envs = setting, N
        for d in range(3):
            c.container[d] = int(container[d] * (1000 if continuous else 1))
        c.internal_node_holder, c.leaf_node_holder, c.env_id_base = I, Lh, base
        self.h = vp()
        ck(L.pct_create(ctypes.byref(c), 0, ctypes.byref(self.h)))
        self.N, self.row = N, (I + Lh + 1) * 9
        ck(L.pct_set_lstsq_mode(self.h, mode))

    def fetch(self):
        HIP.hipDeviceSynchronize()
        obs = np.empty((self.N, self.row), np.float32)
        done = np.empty(self.N, np.uint8)
        flags = np.empty(self.N, np.uint32)
        assert HIP.hipMemcpy(obs.ctypes.data, L.pct_obs(self.h), obs.nbytes, 2) == 0
        assert HIP.hipMemcpy(done.ctypes.data, L.pct_done(self.h), done.nbytes, 2) == 0
        assert HIP.hipMemcpy(flags.ctypes.data, L.pct_error_flags(self.h), flags.nbytes, 2) == 0
        return obs, done, flags

    def reset(self):
        ck(L.pct_reset(self.h, None, 0, None))

    def step(self):
        ck(L.pct_step_hash_policy(self.h, 1, None))


def report(label, ok, detail=""):
    print("%s %s %s" % ("PASS" if ok else "FAIL", label, detail), flush=True)
    return ok


def fixture(name, mode=1):
    c, z = load_case(name)
    env = Env(c["N"], c["setting"], c["container"], c["I"], c["L"], c["base"], mode=mode)
    items = np.ascontiguousarray(np.asarray(case_items(c), np.int32).reshape(-1, 3))
    ck(L.pct_set_item_set(env.h, items.ctypes.data, len(items)))
    st = np.ascontiguousarray(z["stream"].astype(np.int32))
    ck(L.pct_set_item_stream(env.h, st.ctypes.data, st.shape[1]))
    env.reset()
    for t in range(c["steps"] + 1):
        obs, done, flags = env.fetch()
        if not np.array_equal(obs, z["obs"][t]):
            return report(name, False, "observation differs at step %d, envs %s" % (t, np.nonzero((obs != z["obs"][t]).any(1))[0][:8]))
        if t < c["steps"]:
            env.step()
    return report(name, not (flags & ~np.uint32(0xC0)).any(), "%d steps x %d envs == the unmodified reference; notice %s" % (
        c["steps"], c["N"], list((flags & 0x40) != 0)))


def versus_oracle(label, N, steps, okw, make_env, prime):
    ora = OracleVecEnv(N, **okw)
    env = make_env()
    prime(env, ora)
    env.reset()
    ora.reset()
    for t in range(steps):
        obs, done, flags = env.fetch()
        if not np.array_equal(obs, ora.obs.astype(np.float32)):
            return report(label, False, "observation differs at step %d, envs %s" % (t, np.nonzero((obs != ora.obs.astype(np.float32)).any(1))[0][:8]))
        env.step()
        ora.step_hash_policy(1)
    obs, done, flags = env.fetch()
    ok = np.array_equal(obs, ora.obs.astype(np.float32)) and np.array_equal(done, ora.done)
    # (the notice is reported, not compared: which solves of a doomed candidate's virtual check are run differs between the sequential
    # recursion and the wave -- tests/test_zz_gpu_gelsd.py)
    ok = ok and not (flags & ~np.uint32(0xC0)).any()
    if not ok or "-v" in sys.argv:
        oi = ora.ill_conditioned().astype(bool)
        print("   final obs equal %s, done equal %s, error flags %s, notice gpu %s oracle %s" % (
            np.array_equal(obs, ora.obs.astype(np.float32)), np.array_equal(done, ora.done),
            sorted(set(hex(int(f)) for f in flags if f & ~np.uint32(0xC0))), np.nonzero((flags & 0x40) != 0)[0].tolist(), np.nonzero(oi)[0].tolist()),
            flush=True)
    return report(label, ok, "%d envs x %d steps == oracle (gelsd); notices: kernels %d envs, oracle %d" % (
        N, steps, int(((flags & 0x40) != 0).sum()), int(ora.ill_conditioned().astype(bool).sum())))


def main():
    t0 = time.time()
    oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_GELSD)
    if "--wide-only" in sys.argv:  # diagnostic: the wide-flat case alone, in gelsd mode and in the default mode
        items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
        stream = make_stream(4242, 48, 2048, items)[:24]  # (the first 24 envs of the full case, 200 steps: ~1/3 of its ~100 s)
        okw = dict(setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50, env_id_base=5)

        def prime(env, ora):
            a = np.ascontiguousarray(np.asarray(items, np.int32).reshape(-1, 3))
            ck(L.pct_set_item_set(env.h, a.ctypes.data, len(a)))
            s = np.ascontiguousarray(stream.astype(np.int32))
            ck(L.pct_set_item_stream(env.h, s.ctypes.data, s.shape[1]))
            ora.set_item_stream(stream)
        ok = True
        for mode in (1, 0):
            oracle_lib.set_lstsq_mode(mode)
            ok &= versus_oracle("wide_flat 20^3, lstsq mode %d" % mode, 24, 200, okw, lambda: Env(24, 1, (20, 20, 20), 400, 50, 5, mode=mode), prime)
        return 0 if ok else 1
    ok = fixture("discrete_s1_flat_diverging")
    items = [(x, y, 1) for x in range(2, 9) for y in range(2, 9)]
    stream = make_stream(4242, 48, 2048, items)

    def prime_stream(its, st):
        def f(env, ora):
            a = np.ascontiguousarray(np.asarray(its, np.int32).reshape(-1, 3))
            ck(L.pct_set_item_set(env.h, a.ctypes.data, len(a)))
            s = np.ascontiguousarray(st.astype(np.int32))
            ck(L.pct_set_item_stream(env.h, s.ctypes.data, s.shape[1]))
            ora.set_item_stream(st)
        return f
    ok &= versus_oracle("wide_flat 20^3 (up to 16 supporters)", 48, 300,
                        dict(setting=1, container_size=(20, 20, 20), item_set=items, internal_node_holder=400, leaf_node_holder=50, env_id_base=5),
                        lambda: Env(48, 1, (20, 20, 20), 400, 50, 5), prime_stream(items, stream))
    items5 = [(a, b, c) for a in range(1, 6) for b in range(1, 6) for c in range(1, 6)]
    st5 = make_stream(77, 1024, 1024, items5)
    ok &= versus_oracle("C1 domain", 1024, 150,
                        dict(setting=1, container_size=(10, 10, 10), item_set=items5, internal_node_holder=80, leaf_node_holder=50, env_id_base=3),
                        lambda: Env(1024, 1, (10, 10, 10), 80, 50, 3), prime_stream(items5, st5))

    def prime_cont(env, ora):
        ck(L.pct_set_sample_bounds(env.h, 100, 500))
        ck(L.pct_set_sampler(env.h, 21))
        ora.set_sampler(21)
    ok &= versus_oracle("continuous unit bin, setting 1", 256, 120,
                        dict(setting=1, container_size=(1, 1, 1), env_kind=1, sample_bounds=(0.1, 0.5), internal_node_holder=80,
                             leaf_node_holder=50, env_id_base=9),
                        lambda: Env(256, 1, (1, 1, 1), 80, 50, 9, continuous=True), prime_cont)
    ok &= fixture("discrete_s1_flat_lstsq")
    ok &= fixture("discrete_s1_ondomain_avx2", mode=2)  # PCT_LSTSQ_GELSD_AVX2: the reference as it runs on AVX2 hosts
    print("%s in %.0f s" % ("ALL PASS" if ok else "FAILURES", time.time() - t0), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
