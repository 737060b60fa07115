"""On-domain divergence rate of the least-squares stand-in (build container; needs /root/reference).  CPU only.

VERDICT r3 item 5(b): the oracle and the kernels solve the >= 3-supporter split of the stability check with a one-sided
Jacobi SVD where the reference calls np.linalg.lstsq (LAPACK dgelsd).  On adversarial flat-item streams 17 of 56 env-runs
part ways (profiles/r03_lstsq_limit.txt); this script bounds the rate on the reference's OWN item domains with a real
sample: the unmodified reference and the oracle are driven through the same scripted item streams with the stand-in
policy, chunk by chunk in worker processes, and every observation / reward / done / counter is compared.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_lstsq_ondomain.py --procs 6 \
        --discrete-steps 1000000 --continuous-steps 200000 > profiles/r04_lstsq_ondomain.txt
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
    kind, case = job
    from oracle import oracle_lib
    oracle_lib.set_lstsq_mode({"jacobi": oracle_lib.LSTSQ_JACOBI, "gelsd": oracle_lib.LSTSQ_GELSD,
                               "gelsd_avx2": oracle_lib.LSTSQ_GELSD_AVX2}[case.pop("lstsq", "jacobi")])
    g.LSTSQ["calls"] = 0
    if kind == "discrete":
        ref = g.run_reference(case)
        ora = g.run_oracle(case, ref["stream"], ref["density"])
    else:
        ref = g.run_reference_cont(case)
        ora = g.run_oracle_cont(case, ref["stream"], ref["density"])
    calls = g.LSTSQ["calls"]
    first = []
    for e in range(case["N"]):
        bad = np.zeros(case["steps"] + 1, bool)
        bad |= (ref["obs"][:, e] != ora["obs"][:, e]).any(axis=1)
        for k in ("reward", "done", "counter", "ratio"):
            bad[:-1] |= ref[k][:, e] != ora[k][:, e]
        first.append(int(np.argmax(bad)) if bad.any() else -1)
    episodes = int(ref["done"].sum())
    return kind, case["seed"], calls, first, episodes, case["N"] * case["steps"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--discrete-steps", type=int, default=1000000)
    ap.add_argument("--continuous-steps", type=int, default=200000)
    ap.add_argument("--seed0", type=int, default=100000)
    ap.add_argument("--setting", type=int, default=1, choices=[1, 3], help="1: unit densities; 3: per-item densities (scripted)")
    ap.add_argument("--lstsq", default="jacobi", choices=["jacobi", "gelsd", "gelsd_avx2"],
                    help="the oracle's solver: the Jacobi stand-in (the kernels' default) or the dgelsd restatement (pct_oracle_gelsd.c) with "
                         "OpenBLAS' SkylakeX kernel arithmetic (gelsd) / its Haswell one (gelsd_avx2: run the reference with "
                         "OPENBLAS_CORETYPE=HASWELL, or on an AVX2 host)")
    ap.add_argument("--only-discrete-seeds", default="", help="comma-separated chunk seeds: run just these discrete chunks")
    a = ap.parse_args()
    # the C1 domain (configs[0]: setting 1, 10^3, items 1..5, 80 / 50) and the continuous setting-1 unit bin of c3s1
    dcase = dict(setting=a.setting, container=(10, 10, 10), lo=1, hi=5, I=80, L=50, N=4, steps=2000, stream_T=4096, base=0)
    ccase = dict(setting=a.setting, container=(1, 1, 1), lo=0.1, hi=0.5, I=80, L=50, N=2, steps=1000, stream_T=4096, base=0,
                 z_choice=True)
    jobs = []
    nd = -(-a.discrete_steps // (dcase["N"] * dcase["steps"]))
    nc = -(-a.continuous_steps // (ccase["N"] * ccase["steps"]))
    for i in range(max(nd, nc)):  # interleaved, so that a partial log covers both domains
        if i < nd:
            jobs.append(("discrete", dict(dcase, seed=a.seed0 + i, base=7 * i, lstsq=a.lstsq)))
        if i < nc:
            jobs.append(("continuous", dict(ccase, seed=a.seed0 + 50000 + i, base=11 * i, lstsq=a.lstsq)))
    if a.only_discrete_seeds:
        only = [int(x) for x in a.only_discrete_seeds.split(",")]
        jobs = [("discrete", dict(dcase, seed=sd, base=7 * (sd - a.seed0), lstsq=a.lstsq)) for sd in only]
        nd, nc = len(jobs), 0
    print("On-domain sample of the least-squares stand-in: unmodified reference (np.linalg.lstsq = LAPACK dgelsd) vs the C oracle")
    print("(%s), same scripted item streams, stand-in policy, every observation / reward / done / counter / ratio compared."
          % ("one-sided Jacobi SVD" if a.lstsq == "jacobi" else "--lstsq %s: oracle/pct_oracle_gelsd.c, dgelsd operation for operation" % a.lstsq))
    print("OPENBLAS_CORETYPE=%s" % os.environ.get("OPENBLAS_CORETYPE", "(native)"))
    print("discrete: %s\ncontinuous: %s" % (dcase, ccase))
    print("%d + %d chunks on %d processes" % (nd, nc, a.procs), flush=True)
    tot = {k: dict(steps=0, calls=0, runs=0, div=0, episodes=0, same=0) for k in ("discrete", "continuous")}
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        for n, (kind, seed, calls, first, episodes, steps) in enumerate(pool.imap_unordered(chunk, jobs)):
            t = tot[kind]
            t["steps"] += steps
            t["calls"] += calls
            t["runs"] += len(first)
            t["episodes"] += episodes
            case_steps = steps // len(first)
            for f in first:
                t["div"] += f >= 0
                t["same"] += case_steps if f < 0 else f
            if any(f >= 0 for f in first):
                print("  DIVERGED %s seed %d: first differing step per env %s (%d lstsq calls in the chunk)" % (kind, seed, first, calls), flush=True)
            if (n + 1) % 10 == 0 or n + 1 == len(jobs):
                print("  [%6.0f s] %d / %d chunks; discrete %d env-steps, %d lstsq calls, %d / %d env-runs diverged; "
                      "continuous %d env-steps, %d lstsq calls, %d / %d diverged"
                      % (time.time() - t0, n + 1, len(jobs), tot["discrete"]["steps"], tot["discrete"]["calls"],
                         tot["discrete"]["div"], tot["discrete"]["runs"], tot["continuous"]["steps"],
                         tot["continuous"]["calls"], tot["continuous"]["div"], tot["continuous"]["runs"]), flush=True)
    for kind in ("discrete", "continuous"):
        t = tot[kind]
        print("=> %s: %d env-steps in %d env-runs (%d episodes), %d lstsq calls in the reference; %d env-runs parted ways; "
              "%d env-steps identical up to the first difference"
              % (kind, t["steps"], t["runs"], t["episodes"], t["calls"], t["div"], t["same"]))
    print("wall clock %.0f s on %d processes" % (time.time() - t0, a.procs))


if __name__ == "__main__":
    main()
