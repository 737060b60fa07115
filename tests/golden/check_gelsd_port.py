"""Pins oracle/pct_oracle_gelsd.c to the solver the reference actually runs (build container; needs /root/reference for --streams).

np.linalg.lstsq of the reference's stability check (D/space.py:152,249; C/space.py:148,245) is LAPACK dgelsd inside the OpenBLAS that
the NumPy wheel bundles (numpy.libs/libscipy_openblas64_*.so).  That library is loaded here through ctypes (ILP64 symbols
scipy_<name>_64_) and every routine of the restatement is compared BIT FOR BIT with the library's own routine:

  1. BLAS kernels: dnrm2 (x87), dgemv 'T' / 'N', dger, drot (dgemm 'T','N' n x 1: in 3.) -- random sizes up to 40 x 20 / strides
  2. LAPACK auxiliaries: dlartg, dlas2, dlasv2, dlapy2, dlarfg, dbdsqr (with vectors)   -- random inputs incl. graded / split ones
  3. the whole solve against np.linalg.lstsq: systems built the way the reference builds them (k = 3 .. 25 supporters, integer,
     3-decimal and highly degenerate geometry -> rank-deficient systems) and the systems recorded from reference runs
  4. --write-fixture: tests/golden/lstsq_systems.npz (recorded + constructed systems with NumPy's x / rank / singular values), the
     vectors tests/test_gelsd_port.py checks on any machine (lstsq_systems_avx2.npz when run with OPENBLAS_CORETYPE=HASWELL: the
     kernel set of AVX2 hosts, checked against the restatement's AVX2 flavour)
  5. --streams: the unmodified reference against the oracle in gelsd mode on the adversarial flat-item streams of
     check_lstsq_limit.py (17 of 56 env-runs part ways under the Jacobi stand-in) -- every observation compared

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_gelsd_port.py --streams > profiles/r04_gelsd_port.txt
"""
import argparse
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_lib  # noqa: E402

I64, D, I = C.c_int64, C.c_double, C.c_int


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def ip(v):
    return C.byref(I64(v))


def dp(v):
    return C.byref(D(v))


def openblas():
    p = glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so"))
    if not p:
        raise SystemExit("this NumPy does not bundle scipy-openblas64: nothing to pin against")
    L = C.CDLL(p[0])
    g = L.scipy_openblas_get_corename64_
    g.restype = C.c_char_p
    cfg = L.scipy_openblas_get_config64_
    cfg.restype = C.c_char_p
    return L, g().decode(), cfg().decode()


def fn(L, name):
    f = getattr(L, "scipy_" + name + "_64_")
    f.restype = None
    return f


def system(rng, k, mode):
    """coefficient / value of D/space.py:134-151 for k supporter centres and a centre of mass"""
    if mode == 0:
        pts = rng.integers(0, 20, (k, 2)) / 2.0
    elif mode == 1:
        pts = np.round(rng.uniform(0, 1, (k, 2)), 3)
    else:  # many collinear / repeated coordinates: rank-deficient systems, rows dropped by `molecular != 0`
        pts = rng.integers(0, 6, (k, 2)) / 2.0
    com = pts.mean(0) + (rng.standard_normal(2) * 0.3 if mode != 2 else rng.integers(-2, 3, 2) / 4.0)
    M = k * (k - 1) // 2 + 1
    A = np.zeros((M, k))
    b = np.zeros((M, 1))
    r = 0
    for i in range(k - 1):
        for j in range(i + 1, k):
            t = pts[i] - pts[j]
            mol = np.dot(com - pts[i], t)
            if mol != 0:
                A[r, i] = 1
                A[r, j] = -abs(np.dot(com - pts[j], t)) / mol
            r += 1
    A[-1, :] = 1
    b[-1, 0] = 1
    return A, b


def check_blas(L, G, rng, n_trials):
    bad = dict(dnrm2=0, dgemv_t=0, dgemv_n=0, dger=0, drot=0)
    nrm = L.scipy_cblas_dnrm264_
    nrm.restype = D
    G.gelsd_dnrm2.restype = D
    for _ in range(n_trials):
        n = int(rng.integers(1, 130))
        inc = int(rng.integers(1, 4))
        x = rng.standard_normal(n * inc) * 10.0 ** rng.integers(-3, 4)
        bad["dnrm2"] += nrm(I64(n), P(x), I64(inc)) != G.gelsd_dnrm2(I(n), P(x), I(inc))
        # dgemv: leading dimensions >= 4 (dgelsd's lda is M >= 4; OpenBLAS has special cases for lda == m mod 4 <= 3)
        m = int(rng.integers(1, 40))
        n = int(rng.integers(1, 20))
        lda = max(4, m + int(rng.integers(0, 5)))
        A = np.asfortranarray(rng.standard_normal((lda, n)))
        alpha = float(rng.choice([1.0, 1.0, -0.7]))
        for trans, lx, ly, key in (("T", m, n, "dgemv_t"), ("N", n, m, "dgemv_n")):
            incx = int(rng.choice([1, 1, 2, lda]))
            x = rng.standard_normal(lx * incx)
            y = np.zeros(ly)
            fn(L, "dgemv")(C.c_char_p(trans.encode()), ip(m), ip(n), dp(alpha), P(A), ip(lda), P(x), ip(incx), dp(0.0), P(y), ip(1),
                           C.c_size_t(1))
            y2 = np.zeros(ly)
            getattr(G, "gelsd_" + key)(I(m), I(n), D(alpha), P(A), I(lda), P(x), I(incx), P(y2))
            bad[key] += not np.array_equal(y, y2)
        incx, incy = int(rng.choice([1, 2, lda])), int(rng.choice([1, 3]))
        x = rng.standard_normal(m * incx)
        y = rng.standard_normal(n * incy)
        A1, A2 = A.copy(order="F"), A.copy(order="F")
        al = float(rng.standard_normal())
        fn(L, "dger")(ip(m), ip(n), dp(al), P(x), ip(incx), P(y), ip(incy), P(A1), ip(lda))
        G.gelsd_dger(I(m), I(n), D(al), P(x), I(incx), P(y), I(incy), P(A2), I(lda))
        bad["dger"] += not np.array_equal(A1, A2)
        n = int(rng.integers(1, 40))
        inc = int(rng.choice([1, 2, 7]))
        x, y = rng.standard_normal(n * inc), rng.standard_normal(n * inc)
        th = rng.uniform(0, 6.3)
        x2, y2 = x.copy(), y.copy()
        fn(L, "drot")(ip(n), P(x), ip(inc), P(y), ip(inc), dp(np.cos(th)), dp(np.sin(th)))
        G.gelsd_drot(I(n), P(x2), I(inc), P(y2), I(inc), D(np.cos(th)), D(np.sin(th)))
        bad["drot"] += not (np.array_equal(x, x2) and np.array_equal(y, y2))
    return bad


def check_lapack(L, G, rng, n_trials):
    bad = dict(dlartg=0, dlas2=0, dlasv2=0, dlapy2=0, dlarfg=0, dbdsqr=0)
    G.gelsd_dlapy2.restype = D
    G.gelsd_dbdsqr.restype = I
    lp = fn(L, "dlapy2")
    lp.restype = D

    def rnd():
        r = rng.standard_normal() * 10.0 ** rng.integers(-8, 9)
        return float(rng.choice([r, r, r, 0.0, 1.0]))
    for t in range(n_trials):
        f, g, h = rnd(), rnd(), rnd()
        o, o2 = [D() for _ in range(6)], [D() for _ in range(6)]
        fn(L, "dlartg")(dp(f), dp(g), C.byref(o[0]), C.byref(o[1]), C.byref(o[2]))
        G.gelsd_dlartg(D(f), D(g), C.byref(o2[0]), C.byref(o2[1]), C.byref(o2[2]))
        bad["dlartg"] += [v.value for v in o[:3]] != [v.value for v in o2[:3]]
        fn(L, "dlas2")(dp(f), dp(g), dp(h), C.byref(o[0]), C.byref(o[1]))
        G.gelsd_dlas2(D(f), D(g), D(h), C.byref(o2[0]), C.byref(o2[1]))
        bad["dlas2"] += [v.value for v in o[:2]] != [v.value for v in o2[:2]]
        fn(L, "dlasv2")(dp(f), dp(g), dp(h), *[C.byref(v) for v in o])
        G.gelsd_dlasv2(D(f), D(g), D(h), *[C.byref(v) for v in o2])
        bad["dlasv2"] += [v.value for v in o] != [v.value for v in o2]
        bad["dlapy2"] += lp(dp(f), dp(g)) != G.gelsd_dlapy2(D(f), D(g))
        if t % 4:
            continue
        n = int(rng.integers(1, 40))
        inc = int(rng.choice([1, 5]))
        x = rng.standard_normal(n * inc)
        al, tau = D(rng.standard_normal()), D()
        x2, al2, tau2 = x.copy(), D(al.value), D()
        fn(L, "dlarfg")(ip(n), C.byref(al), P(x), ip(inc), C.byref(tau))
        G.gelsd_dlarfg(I(n), C.byref(al2), P(x2), I(inc), C.byref(tau2))
        bad["dlarfg"] += not (np.array_equal(x, x2) and al.value == al2.value and tau.value == tau2.value)
        n = int(rng.integers(1, 26))
        d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 1))
        if t % 3 == 0:
            d = d * 10.0 ** rng.integers(-9, 1, n)
        if t % 5 == 0:
            e[rng.integers(0, len(e))] = 0.0
        vt, c = np.asfortranarray(np.eye(n)), np.asfortranarray(rng.standard_normal((n, 1)))
        d2, e2, vt2, c2 = d.copy(), e.copy(), vt.copy(order="F"), c.copy(order="F")
        work, info, u = np.zeros(4 * n + 8), I64(0), np.zeros(1)
        fn(L, "dbdsqr")(C.c_char_p(b"U"), ip(n), ip(n), ip(0), ip(1), P(d), P(e), P(vt), ip(n), P(u), ip(1), P(c), ip(n), P(work),
                        C.byref(info), C.c_size_t(1))
        i2 = G.gelsd_dbdsqr(I(n), I(n), P(d2), P(e2), P(vt2), I(n), P(c2), P(np.zeros(4 * n + 8)))
        bad["dbdsqr"] += not (np.array_equal(d, d2) and np.array_equal(vt, vt2) and np.array_equal(c, c2) and info.value == i2)
    return bad


def solve_matches(A, b, want=None):
    x, res, rk, sv = want if want is not None else np.linalg.lstsq(A, b, rcond=None)
    x2, rk2, sv2, _ = oracle_lib.gelsd_lstsq(A, b)
    return np.array_equal(np.asarray(x).ravel(), x2) and np.array_equal(sv, sv2) and int(rk) == rk2


def record_systems(seeds, steps):
    """(A, b) of every lstsq call of reference runs on the C1 domain"""
    import gen_golden as g
    rec = []
    inner = np.linalg.lstsq

    def hook(a, b, rcond=None):
        rec.append((np.array(a), np.array(b)))
        return inner(a, b, rcond=rcond)
    np.linalg.lstsq = hook
    try:
        for sd in seeds:
            g.run_reference(dict(setting=1, container=(10, 10, 10), lo=1, hi=5, I=80, L=50, N=2, steps=steps, stream_T=4096,
                                 base=7 * (sd - 100000), seed=sd))
    finally:
        np.linalg.lstsq = inner
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=4000)
    ap.add_argument("--streams", action="store_true", help="also the adversarial flat-item streams against the unmodified reference")
    ap.add_argument("--write-fixture", action="store_true", help="write tests/golden/lstsq_systems.npz")
    a = ap.parse_args()
    L, core, cfg = openblas()
    G = C.CDLL(oracle_lib.build())
    print("NumPy %s; bundled OpenBLAS: %s; kernel set in use: %s" % (np.__version__, cfg.strip(), core))
    if core == "Haswell":  # (OPENBLAS_CORETYPE=HASWELL / ZEN, or an AVX2 host)
        oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_GELSD_AVX2)
        oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_JACOBI)  # (the kernel set stays selected for the direct calls below)
        G.gelsd_set_kernel_set(1)
        print("OpenBLAS runs its Haswell kernel set here: checking the restatement's AVX2 flavour (LSTSQ_GELSD_AVX2)")
    elif core != "SkylakeX":
        print("NOTE: oracle/pct_oracle_gelsd.c restates the SkylakeX and Haswell kernel sets; on this host OpenBLAS runs %s kernels -- "
              "mismatches below are then the reference differing from ITSELF across machines (profiles/r04_lstsq_ondomain.txt)" % core)
    rng = np.random.default_rng(20260925)
    bad = check_blas(L, G, rng, a.trials)
    print("1. BLAS kernels, %d random calls each: mismatches %s" % (a.trials, bad))
    total = sum(bad.values())
    bad = check_lapack(L, G, rng, a.trials * 4)
    print("2. LAPACK auxiliaries, %d (scalar routines) / %d (dlarfg, dbdsqr with vectors) random calls: mismatches %s"
          % (a.trials * 4, a.trials, bad))
    total += sum(bad.values())
    nb = nsys = ndef = 0
    fixture = []
    for t in range(a.trials * 6):
        k = int(rng.choice([3, 3, 4, 4, 5, 6, 7, 8, 10, 12, 16, 20, 25]))
        A, b = system(rng, k, t % 3)
        want = np.linalg.lstsq(A, b, rcond=None)
        nsys += 1
        ndef += int(want[2]) < k
        nb += not solve_matches(A, b, want)
        if t < 900 and k <= 16:
            fixture.append((A, b, want))
    print("3a. np.linalg.lstsq on %d constructed systems (k = 3 .. 25; %d rank-deficient): mismatches (x, rank or singular values) %d"
          % (nsys, ndef, nb))
    total += nb
    if os.path.isdir("/root/reference"):
        rec = record_systems([100014, 100046], 1000)
        nb = 0
        for A, b in rec:
            want = np.linalg.lstsq(A, b, rcond=None)
            nb += not solve_matches(A, b, want)
        print("3b. %d systems recorded from reference runs (C1 domain): mismatches %d" % (len(rec), nb))
        total += nb
        if a.write_fixture:
            seen = set()
            for A, b in rec:
                key = A.tobytes()
                if key not in seen and len(seen) < 600:
                    seen.add(key)
                    fixture.append((A, b, np.linalg.lstsq(A, b, rcond=None)))
    if a.write_fixture:
        kmax = 16
        n = len(fixture)
        Ms = np.array([f[0].shape[0] for f in fixture], np.int32)
        Ns = np.array([f[0].shape[1] for f in fixture], np.int32)
        Aflat = np.concatenate([f[0].ravel() for f in fixture])
        X = np.zeros((n, kmax))
        S = np.zeros((n, kmax))
        R = np.zeros(n, np.int32)
        for i, (A, b, w) in enumerate(fixture):
            X[i, :Ns[i]] = np.asarray(w[0]).ravel()
            S[i, :Ns[i]] = w[3]
            R[i] = int(w[2])
        fname = "lstsq_systems_avx2.npz" if core == "Haswell" else "lstsq_systems.npz"
        np.savez_compressed(os.path.join(HERE, fname), M=Ms, N=Ns, A=Aflat, x=X, sv=S, rank=R,
                            meta=np.array("np.linalg.lstsq(A, e_M, rcond=None) of NumPy %s (%s, %s kernels); A row-major, b = last unit "
                                          "vector" % (np.__version__, cfg.strip(), core)))
        print("4. wrote tests/golden/%s: %d systems" % (fname, n))
    if a.streams:
        import gen_golden as g
        strict = oracle_lib.LSTSQ_GELSD_AVX2 if core == "Haswell" else oracle_lib.LSTSQ_GELSD
        for label, mode in (("Jacobi stand-in", oracle_lib.LSTSQ_JACOBI), ("gelsd restatement (%s kernel set)" % core, strict)):
            oracle_lib.set_lstsq_mode(mode)
            runs = div = steps = calls = 0
            for name in ("discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq"):
                for seed in range(61, 69):
                    case = dict(g.CASES[name], seed=seed)
                    g.LSTSQ["calls"] = 0
                    ref = g.run_reference(case)
                    ora = g.run_oracle(case, ref["stream"], ref["density"])
                    badm = np.argwhere(ref["obs"] != ora["obs"])
                    per_env = [int(badm[badm[:, 1] == e][:, 0].min()) if (badm[:, 1] == e).any() else -1 for e in range(case["N"])]
                    runs += case["N"]
                    div += sum(1 for x in per_env if x >= 0)
                    steps += sum((x if x >= 0 else case["steps"]) for x in per_env)
                    calls += g.LSTSQ["calls"]
            print("5. adversarial flat-item streams, unmodified reference vs oracle with the %s: %d env-runs, %d parted ways, "
                  "%d env-steps identical, %d lstsq calls" % (label, runs, div, steps, calls), flush=True)
            if mode == strict:
                total += div
        oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_JACOBI)
    print("TOTAL mismatches: %d" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
