"""Fixtures of the unmodified reference as it runs on an AVX2 host (OpenBLAS' "Haswell" kernel set: Intel Haswell .. / AMD Zen).

    OPENBLAS_CORETYPE=HASWELL PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_avx2.py      (build container; needs /root/reference)

NumPy's bundled OpenBLAS picks its kernels by CPU; OPENBLAS_CORETYPE forces the set an AVX2 host would get.  With it np.dot of two
2-vectors is x0*y0 + x1*y1 (no FMA), and dgemv 'N' / daxpy / dgemm inside np.linalg.lstsq sum differently: the reference's stability check
(settings 1 / 3) then takes another path at ties than on an AVX-512 host (profiles/r04_lstsq_ondomain.txt).  The oracle's and the kernels'
PCT_LSTSQ_GELSD_AVX2 flavour restates that arithmetic; each fixture is written only if the oracle in that mode equals the reference.
  discrete_s1_ondomain_avx2   the C1 domain, chunk seed 100076: the AVX-512 and the AVX2 reference part ways at step 115 of env 1
                              (`first_difference_from_avx512`: where the oracle in PCT_LSTSQ_GELSD mode leaves this recording)
  discrete_s1_flat_lstsq_avx2 the adversarial flat-item stream of discrete_s1_flat_lstsq
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as g  # noqa: E402
import check_gelsd_port as cg  # noqa: E402
from oracle import oracle_lib  # noqa: E402


def main():
    _, core, cfg = cg.openblas()
    if core != "Haswell":
        raise SystemExit("run with OPENBLAS_CORETYPE=HASWELL (OpenBLAS reports %s)" % core)
    cases = {
        "discrete_s1_ondomain_avx2": dict(setting=1, container=(10, 10, 10), lo=1, hi=5, I=80, L=50, N=4, steps=200, stream_T=4096,
                                          base=7 * 76, seed=100076),
        "discrete_s1_flat_lstsq_avx2": dict(g.CASES["discrete_s1_flat_lstsq"]),
    }
    for name, case in cases.items():
        g.LSTSQ["calls"] = 0
        ref = g.run_reference(case)
        case = dict(case, lstsq_calls=g.LSTSQ["calls"], openblas=cfg.strip())
        first = {}
        for label, mode in (("avx2", oracle_lib.LSTSQ_GELSD_AVX2), ("avx512", oracle_lib.LSTSQ_GELSD)):
            oracle_lib.set_lstsq_mode(mode)
            ora = g.run_oracle(case, ref["stream"], ref["density"])
            bad = np.argwhere((ref["obs"] != ora["obs"]).any(2))
            first[label] = [int(bad[bad[:, 1] == e][:, 0].min()) if (bad[:, 1] == e).any() else -1 for e in range(case["N"])]
            if label == "avx2":
                for key in ("obs", "reward", "done", "counter"):
                    if not np.array_equal(ref[key], ora[key]):
                        raise SystemExit("MISMATCH %s/%s" % (name, key))
        oracle_lib.set_lstsq_mode(oracle_lib.LSTSQ_JACOBI)
        print("%-30s steps=%d envs=%d lstsq calls=%d  oracle (GELSD_AVX2) == reference on %s kernels; the GELSD (AVX-512) flavour leaves it at %s"
              % (name, case["steps"], case["N"], case["lstsq_calls"], core, first["avx512"]))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=np.array(repr(case)), stream=ref["stream"], obs=ref["obs"],
                            reward=ref["reward"], done=ref["done"], counter=ref["counter"], ratio=ref["ratio"] * (ref["done"] != 0),
                            first_difference_from_avx512=np.array(first["avx512"]))


if __name__ == "__main__":
    main()
