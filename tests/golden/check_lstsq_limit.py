"""How far the >= 3-supporter split of the stability check can be pinned (build container; needs /root/reference).

The reference solves that split with np.linalg.lstsq = LAPACK dgelsd inside NumPy's OpenBLAS; the oracle and the
kernels run a one-sided Jacobi SVD of A (the same minimum-norm solution; the last bits differ).
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
    """oracle/pct_oracle_stab.c lstsq_min_norm (one-sided Jacobi SVD), operation for operation"""
    A = np.asarray(A, float)
    b = np.asarray(b, float).reshape(-1)
    M, N = A.shape
    U, V = A.copy(), np.eye(N)
    eps = 2.220446049250313e-16
    for _ in range(60):
        rotated = False
        for p in range(N):
            for q in range(p + 1, N):
                alpha = beta = gamma = 0.0
                for r in range(M):
                    alpha += U[r, p] * U[r, p]
                    beta += U[r, q] * U[r, q]
                    gamma += U[r, p] * U[r, q]
                if gamma == 0 or abs(gamma) <= eps * math.sqrt(alpha * beta):
                    continue
                rotated = True
                zeta = (beta - alpha) / (2 * gamma)
                t = (1.0 if zeta >= 0 else -1.0) / (abs(zeta) + math.sqrt(1 + zeta * zeta))
                c = 1 / math.sqrt(1 + t * t)
                sn = c * t
                for r in range(M):
                    up, uq = U[r, p], U[r, q]
                    U[r, p] = c * up - sn * uq
                    U[r, q] = sn * up + c * uq
                for r in range(N):
                    vp, vq = V[r, p], V[r, q]
                    V[r, p] = c * vp - sn * vq
                    V[r, q] = sn * vp + c * vq
        if not rotated:
            break
    s2 = []
    for j in range(N):
        a2 = 0.0
        for r in range(M):
            a2 += U[r, j] * U[r, j]
        s2.append(a2)
    smax2 = max(s2)
    rc = eps * max(M, N)
    x = np.zeros(N)
    for j in range(N):
        if s2[j] <= 0 or math.sqrt(s2[j]) <= rc * math.sqrt(smax2):
            continue
        proj = 0.0
        for r in range(M):
            proj += U[r, j] * b[r]
        proj /= s2[j]
        for i in range(N):
            x[i] += V[i, j] * proj
    return x.reshape(-1, 1)


def main():
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


if __name__ == "__main__":
    main()
