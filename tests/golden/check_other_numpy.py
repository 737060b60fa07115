"""Does ANOTHER NumPy release give the recorded np.linalg.lstsq solutions?  (build container: the image holds a second Python,
/opt/conda/bin/python3.9 with NumPy 1.26.4 = OpenBLAS 0.3.23.dev; run this file WITH THAT INTERPRETER, no other dependency)

    env -i PATH=/opt/conda/bin:/usr/bin:/bin OPENBLAS_CORETYPE=SKYLAKEX /opt/conda/bin/python3.9 tests/golden/check_other_numpy.py \
        tests/golden/lstsq_systems.npz tests/golden/lstsq_systems_avx2.npz
    (and OPENBLAS_CORETYPE=HASWELL; unset, that OpenBLAS falls back to its generic "Prescott" kernels on this virtual CPU)

tests/golden/lstsq_systems*.npz were recorded with NumPy 2.2.6 / OpenBLAS 0.3.29 (tests/golden/check_gelsd_port.py).  Whole reference
runs under the other NumPy: tests/golden/check_lstsq_ondomain.py with the same interpreter (ref_shim.py stubs the torch import it lacks).
Results: profiles/r04_gelsd_other_numpy.txt."""
import numpy as np, sys, ctypes, glob, os
lib = glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libopenblas*.so"))[0]
L = ctypes.CDLL(lib)
core = None
for sym in ("openblas_get_corename64_", "openblas_get_corename"):
    if hasattr(L, sym):
        f = getattr(L, sym); f.restype = ctypes.c_char_p; core = f().decode(); break
cfg = None
for sym in ("openblas_get_config64_", "openblas_get_config"):
    if hasattr(L, sym):
        f = getattr(L, sym); f.restype = ctypes.c_char_p; cfg = f().decode(); break
print("NumPy", np.__version__, "|", cfg, "| kernel set:", core)
for fname in sys.argv[1:]:
    z = np.load(fname)
    off = 0; same = rank_same = n = 0; byk = {}
    for i in range(len(z["M"])):
        m, k = int(z["M"][i]), int(z["N"][i])
        a = z["A"][off:off + m * k].reshape(m, k); off += m * k
        b = np.zeros(m); b[-1] = 1.0
        x, res, rk, sv = np.linalg.lstsq(a, b, rcond=None)
        ok = np.array_equal(x, z["x"][i, :k]) and np.array_equal(sv, z["sv"][i, :k]) and int(rk) == int(z["rank"][i])
        same += ok; n += 1
        t = byk.setdefault(k, [0, 0]); t[0] += ok; t[1] += 1
    print(os.path.basename(fname), ": identical (x, rank, singular values) on", same, "of", n, {k: tuple(v) for k, v in sorted(byk.items())})
