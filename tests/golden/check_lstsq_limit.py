"""How far the >= 3-supporter split of the stability check can be pinned (build container; needs /root/reference).

The reference solves that split with np.linalg.lstsq = LAPACK dgelsd inside NumPy's OpenBLAS; the oracle and the
kernels run a Jacobi eigen-solve of A^T A (same minimum-norm solution, last bits differ: max |difference| ~ 4e-15).
With integer geometry the tests downstream are often EXACTLY degenerate (a stack centre on the line through a
polygon edge -> point_in_polygen returns False on `cross == 0`, convex_hull.py:104-105), so that last bit decides
placements.  This script measures it on the adversarial "flat" item sets of gen_golden.py (every item of height 1:
tops align, wide boxes rest on 3+ supporters) over the stream seeds 61..68:

  A. unmodified reference vs oracle            -> how often a run parts ways, after how many lstsq calls;
  B. reference with np.linalg.lstsq REPLACED by a Python port of the oracle's Jacobi solve vs oracle
                                               -> must be identical on every run: then the solver's last bit is the
                                                  ONLY difference left (everything else -- traversal order, the
                                                  aliasing of thisStack objects, NumPy's FMA dot products -- is exact).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_lstsq_limit.py > profiles/r02_lstsq_limit.txt
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as g  # noqa: E402


def jacobi_lstsq(A, b):
    """oracle/pct_oracle_stab.c lstsq_min_norm, operation for operation"""
    A = np.asarray(A, float)
    b = np.asarray(b, float).reshape(-1)
    M, N = A.shape
    G, V, gg = np.zeros((N, N)), np.eye(N), np.zeros(N)
    for i in range(N):
        s = 0.0
        for r in range(M):
            s += A[r, i] * b[r]
        gg[i] = s
        for j in range(N):
            s = 0.0
            for r in range(M):
                s += A[r, i] * A[r, j]
            G[i, j] = s
    for _ in range(60):
        off = 0.0
        for p in range(N):
            for q in range(p + 1, N):
                off += G[p, q] * G[p, q]
        if off < 1e-300:
            break
        for p in range(N):
            for q in range(p + 1, N):
                if abs(G[p, q]) < 1e-300:
                    continue
                theta = (G[q, q] - G[p, p]) / (2 * G[p, q])
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + math.sqrt(theta * theta + 1))
                c = 1 / math.sqrt(t * t + 1)
                sn = t * c
                for k in range(N):
                    gkp, gkq = G[k, p], G[k, q]
                    G[k, p] = c * gkp - sn * gkq
                    G[k, q] = sn * gkp + c * gkq
                for k in range(N):
                    gpk, gqk = G[p, k], G[q, k]
                    G[p, k] = c * gpk - sn * gqk
                    G[q, k] = sn * gpk + c * gqk
                for k in range(N):
                    vkp, vkq = V[k, p], V[k, q]
                    V[k, p] = c * vkp - sn * vkq
                    V[k, q] = sn * vkp + c * vkq
    smax = max(G[i, i] for i in range(N))
    rc = 2.220446049250313e-16 * max(M, N)
    x = np.zeros(N)
    for k in range(N):
        lam = G[k, k]
        if lam <= 0 or math.sqrt(lam) <= rc * math.sqrt(smax):
            continue
        proj = 0.0
        for i in range(N):
            proj += V[i, k] * gg[i]
        proj /= lam
        for i in range(N):
            x[i] += V[i, k] * proj
    return x.reshape(-1, 1)


MODE = {"jacobi": False, "maxdiff": 0.0}
_counting = np.linalg.lstsq  # gen_golden's counting wrapper around LAPACK


def switchable(A, b, rcond=None):
    r = _counting(A, b, rcond=rcond)
    x2 = jacobi_lstsq(A, b)
    MODE["maxdiff"] = max(MODE["maxdiff"], float(np.abs(r[0] - x2).max()))
    return ((x2,) + tuple(r[1:])) if MODE["jacobi"] else r


np.linalg.lstsq = switchable
for label, jac in (("A. unmodified reference (LAPACK gelsd) vs oracle", False),
                   ("B. reference with the oracle's Jacobi solve patched in vs oracle", True)):
    MODE["jacobi"] = jac
    print(label)
    runs = div = steps = calls = 0
    for name in ("discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq"):
        for seed in range(61, 69):
            case = dict(g.CASES[name], seed=seed)
            g.LSTSQ["calls"] = 0
            ref = g.run_reference(case)
            ora = g.run_oracle(case, ref["stream"], ref["density"])
            bad = np.argwhere(ref["obs"] != ora["obs"])
            per_env = [int(bad[bad[:, 1] == e][:, 0].min()) if (bad[:, 1] == e).any() else -1 for e in range(case["N"])]
            runs += case["N"]
            div += sum(1 for x in per_env if x >= 0)
            steps += sum((x if x >= 0 else case["steps"]) for x in per_env)
            calls += g.LSTSQ["calls"]
            print("  %-24s seed %d: %5d lstsq calls, first divergence per env %s" % (name, seed, g.LSTSQ["calls"], per_env), flush=True)
    print("  => %d env-runs, %d parted ways, %d env-steps identical, %d lstsq calls; max |gelsd - Jacobi| over all solves %.2e"
          % (runs, div, steps, calls, MODE["maxdiff"]), flush=True)
