"""Builds the HIP C-ABI library in-tree: online-3d-bpp-pct_amd/libpct_hip.so (gfx950).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpct_hip.so")
# (slowest translation units first: they are handed to the worker pool in this order)
SOURCES = ["pct_discrete_stab.hip", "pct_discrete_stab_mt.hip", "pct_discrete_u64_stab.hip", "pct_discrete_u64_stab_mt.hip",
           "pct_continuous.hip", "pct_continuous_mt.hip", "pct_continuous_pipe.hip", "pct_discrete.hip", "pct_discrete_mt.hip", "pct_discrete_u64.hip",
           "pct_discrete_u64_mt.hip", "pct_env.hip"]


def _includes(path, seen=None):
    """the csrc files `path` includes, transitively (quoted includes only: they are the tree's own)"""
    import re
    seen = set() if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
        f = os.path.join(CSRC, name)
        if os.path.exists(f) and f not in seen:
            seen.add(f)
            _includes(f, seen)
    return seen


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 library")


def _tu_inputs(src):
    # what a translation unit is compiled from: a change elsewhere does not recompile it
    return [os.path.join(CSRC, src)] + sorted(_includes(os.path.join(CSRC, src))) + [
        os.path.join(HERE, "..", "include", "pct_env.h"), os.path.abspath(__file__)]  # (this file holds the flags)


def _obj(src):
    return os.path.join(HERE, "build", os.path.splitext(src)[0] + ".o")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for src in SOURCES for d in _tu_inputs(src)):
        return True
    # a source edited while a build was running is older than the library that build linked, but newer than the object
    # it was compiled into (objects stay in the build container; where there are none the library's own time decides)
    for src in SOURCES:
        if os.path.exists(_obj(src)) and any(os.path.getmtime(d) > os.path.getmtime(_obj(src)) for d in _tu_inputs(src)):
            return True
    return False


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    # one hipcc per translation unit, in parallel, then a link step
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
             # the continuous env mirrors the reference's float64 operation order: no FMA contraction
             "-ffp-contract=off",
             # no real calls inside a kernel: every device function is inlined (profiles/r03_fault_root_cause.txt: this
             # hipcc mis-places AGPR split copies around a call under a narrowed exec mask); scripts/check_no_calls.py
             # verifies the built library
             "-mllvm", "-amdgpu-function-calls=false"]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = _obj(src)
        srcs = _tu_inputs(src)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in srcs):
            return obj
        cmd = [_hipcc(), *flags, "-Wno-pass-failed", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        t0 = time.time()
        subprocess.check_call(cmd)
        if verbose:
            print("%s: %.0f s" % (src, time.time() - t0))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
