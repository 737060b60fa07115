"""Is the reference's OWN trajectory independent of the BLAS kernels its NumPy runs on?  (build container; needs /root/reference)

tests/golden/check_lstsq_ondomain.py finds about one env-run in fifty (2000 steps each, the reference's own item domain,
discrete setting 1) where the unmodified reference and the oracle part ways -- on a last bit of an np.linalg.lstsq
result (LAPACK dgelsd inside NumPy's OpenBLAS) that decides an exactly degenerate point-in-polygon test downstream.  This
script re-runs the reference on those chunks with OpenBLAS forced onto the kernels of other x86 cores
(OPENBLAS_CORETYPE, read when the library is loaded: one subprocess per core type) and reports where each of them
leaves the native run.  A core type that leaves it shows that the reference's verdicts at those steps are a property of
the machine it runs on, not of its algorithm: there is no machine-independent answer to be bit-exact against.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_lstsq_blas.py <ondomain log> [--procs 6] >> profiles/r04_lstsq_ondomain.txt
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CORES = ["native", "SKYLAKEX", "HASWELL", "ZEN", "SANDYBRIDGE", "NEHALEM"]


def worker(seed, base, steps):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import gen_golden as g
    from threadpoolctl import threadpool_info
    case = dict(setting=1, container=(10, 10, 10), lo=1, hi=5, I=80, L=50, N=4, steps=steps, stream_T=4096, base=base, seed=seed)
    ref = g.run_reference(case)
    ora = g.run_oracle(case, ref["stream"], ref["density"])
    arch = [i.get("architecture") for i in threadpool_info() if i.get("internal_api") == "openblas"]
    out = {"arch": arch, "digest": [], "vs_oracle": []}
    for e in range(case["N"]):
        out["digest"].append([hashlib.md5(ref["obs"][t, e].tobytes()).hexdigest()[:12] for t in range(steps + 1)])
        bad = (ref["obs"][:, e] != ora["obs"][:, e]).any(axis=1)
        out["vs_oracle"].append(int(np.argmax(bad)) if bad.any() else -1)
    print("RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--seed0", type=int, default=100000)
    a = ap.parse_args()
    div = []
    for ln in open(a.log):
        m = re.search(r"DIVERGED discrete seed (\d+): first differing step per env \[([^\]]*)\]", ln)
        if m:
            div.append((int(m.group(1)), [int(x) for x in m.group(2).split(",")]))
    print("\nThe reference against ITSELF on other BLAS kernels (tests/golden/check_lstsq_blas.py): the %d diverging discrete chunks of the"
          % len(div))
    print("on-domain sample above, re-run with OPENBLAS_CORETYPE forced (NumPy %s, its bundled OpenBLAS); per env the first step whose"
          % np.__version__)
    print("observation differs from the native run's (-1: identical over the 2000 steps), and the oracle's first difference from THAT run")
    jobs = [(seed, core) for seed, _ in div for core in CORES]
    procs, results = [], {}

    def launch(seed, core):
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        if core != "native":
            env["OPENBLAS_CORETYPE"] = core
        base = 7 * (seed - a.seed0)
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(seed), str(base), "2000"],
                                env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    pending = list(jobs)
    while pending or procs:
        while pending and len(procs) < a.procs:
            j = pending.pop(0)
            procs.append((j, launch(*j)))
        j, p = procs.pop(0)
        out = p.communicate()[0]
        for ln in out.splitlines():
            if ln.startswith("RESULT "):
                results[j] = json.loads(ln[7:])
    n_self = 0
    for seed, first in div:
        nat = results.get((seed, "native"))
        print("  seed %d (oracle vs native reference: %s)" % (seed, first))
        for core in CORES[1:]:
            r = results.get((seed, core))
            if not r or not nat:
                print("    %-12s no result" % core)
                continue
            fd = []
            for e in range(len(nat["digest"])):
                d = [t for t in range(len(nat["digest"][e])) if nat["digest"][e][t] != r["digest"][e][t]]
                fd.append(d[0] if d else -1)
            n_self += any(x >= 0 for x in fd)
            print("    %-12s (OpenBLAS reports %s): leaves the native run at %s; oracle vs this run: %s" % (
                core, ",".join(map(str, r["arch"])), fd, r["vs_oracle"]))
    print("=> %d of %d (chunk, core type) re-runs of the UNMODIFIED reference leave its own native-kernel trajectory (SKYLAKEX is the "
          "native choice of the build container: those re-runs are controls)" % (n_self, len(div) * (len(CORES) - 1)))


if __name__ == "__main__":
    main()
