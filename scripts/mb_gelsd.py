#!/usr/bin/env python
"""Device check + microbenchmark of the strict least-squares split (csrc/pct_gelsd.cuh) through scripts/mb/libmb_gelsd.so.

Systems are built as the stability check builds them (contact centres on a half-unit grid, so that every dot product is exact in
double on any machine); the expected fractions come from the oracle's independent dgelsd restatement (oracle/pct_oracle_gelsd.c,
pinned to NumPy by tests/test_gelsd_port.py) fed the same system.  Prints, per (k, lane-group width G, systems per wave): bit-exact
count and the shader-clock cycles of a solve."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib as ol  # noqa: E402  (the checker)


def make_systems(k, count, rng, narrow_every=2):
    ins = np.zeros((count, 36))
    xs = np.zeros((count, 16))
    near = np.zeros(count, np.int32)
    for t in range(count):
        pts = rng.integers(0, 40 if t % narrow_every else 6, (k, 2)) / 2.0
        com = rng.integers(0, 80 if t % narrow_every else 12, 2) / 4.0
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
        x, rank, sv, nr = ol.gelsd_lstsq(A, b)
        ins[t, 0] = k
        ins[t, 1:3] = com
        ins[t, 4:4 + 2 * k] = pts.ravel()
        xs[t, :k] = x
        near[t] = nr
    return ins, xs, near


def run(lib, ins, per_wave, G, n, variant, avx2=0):
    nsys = len(ins)
    x = np.zeros((nsys, 16))
    ill = np.zeros(nsys, np.int32)
    nblk = (nsys + per_wave - 1) // per_wave
    cyc = np.zeros(nblk, np.uint64)
    rc = lib.mb_gelsd_run(ins.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p), ill.ctypes.data_as(ctypes.c_void_p),
                          cyc.ctypes.data_as(ctypes.c_void_p), nsys, per_wave, G, n, avx2, variant)
    assert rc == 0, rc
    return x, ill, cyc


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "mb", "libmb_gelsd.so"))
    rng = np.random.default_rng(2025)
    quick = "--quick" in sys.argv
    only = os.environ.get("MB_ONLY")  # "k,variant,G,per_wave": one configuration (for a profiler run)
    only = tuple(int(v) for v in only.split(",")) if only else None
    bad = 0
    for k, n, groups in ((3, 4, (1, 4)), (4, 4, (1, 4)), (5, 8, (1, 8)), (8, 8, (1, 8, 16)), (11, 16, (1, 16)), (16, 16, (1, 16, 64))):
        count = 256 if k <= 8 else 64
        if quick:
            count //= 4
        if only and only[0] != k:
            continue
        ins, xs, near = make_systems(k, count, rng)
        for variant in (1,):
            for G in groups:
                for per_wave in sorted({1, max(1, min(64 // G, 8)), 64 // G}):
                    if only and only != (k, variant, G, per_wave):
                        continue
                    if per_wave * (4 + 3 * n + (n * (n - 1) // 2 + 1) * (n + 1) + n * n + 8 * n) * 8 > 150000:
                        continue
                    x, ill, cyc = run(lib, ins, per_wave, G, n, variant)
                    ok = int(np.sum(np.all(x == xs, axis=1)))
                    okn = int(np.sum(ill == near))
                    bad += (count - ok) + (count - okn)
                    print("k=%2d class=%2d %s G=%2d systems/wave=%2d: bit-exact %d/%d, notice %d/%d, cycles/solve-call median %8d  max %8d"
                          % (k, n, ("serial", "group ")[variant], G, per_wave, ok, count, okn, count, int(np.median(cyc)), int(cyc.max())),
                          flush=True)
    print("MISMATCHES", bad)


if __name__ == "__main__":
    main()
