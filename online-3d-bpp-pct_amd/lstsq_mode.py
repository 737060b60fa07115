"""Which least-squares flavour reproduces a reference run on THIS host (pct_set_lstsq_mode, include/pct_env.h).

The reference's stability check calls np.linalg.lstsq and np.dot (D/space.py:110-115,143-152); both end in the OpenBLAS that the NumPy wheel
bundles, which picks its kernels by CPU family at load time.  PCT_LSTSQ_GELSD restates the kernel set of AVX-512 hosts ("SkylakeX"),
PCT_LSTSQ_GELSD_AVX2 the one of AVX2 hosts ("Haswell", which AMD Zen 1-3 also get).  `numpy_lstsq_mode()` asks the NumPy of the running
process which set it uses -- the answer for a reference that runs (or ran) in this Python on this machine."""
import ctypes
import glob
import os

RESTATED = {"SkylakeX": "gelsd", "Haswell": "gelsd_avx2"}
# what the restatement was pinned against (tests/golden/check_gelsd_port.py); other releases may order their sums differently
PINNED_OPENBLAS = "0.3.29"


def _openblas_version(L):
    """the release string out of openblas_get_config ("OpenBLAS 0.3.29 DYNAMIC_ARCH ..."), or None"""
    import re
    for sym in ("scipy_openblas_get_config64_", "scipy_openblas_get_config", "openblas_get_config64_", "openblas_get_config"):
        if hasattr(L, sym):
            f = getattr(L, sym)
            f.restype = ctypes.c_char_p
            m = re.search(r"OpenBLAS\s+(\d+\.\d+\.\d+)", (f() or b"").decode(errors="replace"))
            if m:
                return m.group(1)
    return None


def numpy_blas():
    """(kernel set, OpenBLAS version) of the BLAS behind this process's NumPy, or (None, None) when it cannot be told"""
    try:
        import threadpoolctl
        for lib in threadpoolctl.threadpool_info():
            if lib.get("internal_api") == "openblas" and "numpy" in lib.get("filepath", ""):
                return lib.get("architecture"), lib.get("version")
    except Exception:
        pass
    try:
        import numpy as np
        cand = glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas*.so*"))
        if cand:
            L = ctypes.CDLL(cand[0])
            for sym in ("scipy_openblas_get_corename64_", "scipy_openblas_get_corename"):
                if hasattr(L, sym):
                    f = getattr(L, sym)
                    f.restype = ctypes.c_char_p
                    return f().decode(), _openblas_version(L)
    except Exception:
        pass
    return None, None


# releases whose routines were checked against the restatement (tests/golden/check_gelsd_port.py, check_other_numpy.py: NumPy 1.26.4 ... 2.2.6)
CHECKED_OPENBLAS = ("0.3.23", "0.3.29")


def numpy_lstsq_mode(strict=False):
    """'gelsd' / 'gelsd_avx2' for the kernel set this process's NumPy runs, else None (strict: raise).  An OpenBLAS release outside the
    checked span is a warning -- the mode is still returned -- and an error under strict (a version that cannot be read at all stays a warning):
    another release may order its kernel sums differently, and the claim "bit-identical to this NumPy" would be unfounded."""
    arch, version = numpy_blas()
    mode = RESTATED.get(arch)
    if mode is None:
        if strict:
            raise RuntimeError("NumPy's BLAS on this host runs the %r kernel set; restated are %s" % (arch, sorted(RESTATED)))
        return None
    v = (version or "").split(".dev")[0]
    lo, hi = (tuple(int(x) for x in c.split(".")) for c in CHECKED_OPENBLAS)
    try:
        known = lo <= tuple(int(x) for x in v.split(".")[:3]) <= hi
    except ValueError:
        known = False
    if not known:
        msg = ("NumPy's OpenBLAS is %s; the dgelsd restatement was checked against %s ... %s (pinned: %s)"
               % (version or "of unknown version", CHECKED_OPENBLAS[0], CHECKED_OPENBLAS[1], PINNED_OPENBLAS))
        if strict and v:  # (a version that cannot be READ is a warning, ADVICE r5: a host without threadpoolctl must not lose the mode)
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg)
    return mode


__all__ = ["numpy_blas", "numpy_lstsq_mode", "RESTATED", "PINNED_OPENBLAS", "CHECKED_OPENBLAS"]
