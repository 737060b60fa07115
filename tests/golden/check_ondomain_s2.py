"""Setting 2 (no stability check: integer / float64-lattice work only) on the reference's own item domains, a larger sample than
the fixtures: the unmodified reference against the oracle under every leaf-node expansion scheme (build container; needs
/root/reference; CPU only).  Expected and required: no difference at all.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_ondomain_s2.py --procs 7 --steps 200000 > profiles/r04_ondomain_s2.txt
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def chunk(job):
    import gen_golden as g
    label, kind, case = job
    if kind == "discrete":
        ref = g.run_reference(case)
        ora = g.run_oracle(case, ref["stream"], ref["density"])
    else:
        ref = g.run_reference_cont(case)
        ora = g.run_oracle_cont(case, ref["stream"], ref["density"])
    bad = 0
    for e in range(case["N"]):
        b = (ref["obs"][:, e] != ora["obs"][:, e]).any(axis=1)
        for k in ("reward", "done", "counter", "ratio"):
            b[:-1] |= ref[k][:, e] != ora[k][:, e]
        bad += int(b.any())
    return label, case["N"] * case["steps"], int(ref["done"].sum()), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=7)
    ap.add_argument("--steps", type=int, default=200000, help="env-steps per configuration")
    a = ap.parse_args()
    jobs = []
    for li, lnes in enumerate(["EMS", "EV", "EP", "CP", "FC"]):
        n = -(-a.steps // 8000)
        for i in range(n):
            c = dict(setting=2, container=(10, 10, 10), lo=1, hi=5, I=80, L=50, N=4, steps=2000, stream_T=4096, base=13 * i,
                     seed=500000 + 1000 * li + i)
            if lnes != "EMS":
                c["lnes"] = lnes
            jobs.append(("discrete 10^3 items 1..5 LNES=" + lnes, "discrete", c))
    for i in range(-(-a.steps // 4000)):
        jobs.append(("continuous 10^3 items U(1,5)", "continuous",
                     dict(setting=2, container=(10, 10, 10), lo=1.0, hi=5.0, I=80, L=50, N=2, steps=2000, stream_T=4096, base=17 * i,
                          seed=560000 + i)))
    print("Setting 2 on the reference's own item domains: unmodified reference vs the C oracle, every observation / reward / done /")
    print("counter / ratio compared; %d chunks on %d processes" % (len(jobs), a.procs), flush=True)
    tot = {}
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        for label, steps, episodes, bad in pool.imap_unordered(chunk, jobs):
            t = tot.setdefault(label, [0, 0, 0])
            t[0] += steps
            t[1] += episodes
            t[2] += bad
            if bad:
                print("  DIFFERENCE in", label, flush=True)
    for label in sorted(tot):
        print("=> %-40s %8d env-steps, %6d episodes: %d env-runs differ" % (label, *tot[label]))
    print("wall clock %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
