#!/usr/bin/env python
"""Kernel experiments: builds online-3d-bpp-pct_amd/build/v/lib<name>.so from the library's objects with the given
translation units recompiled under extra -D flags (select it at run time with PCT_HIP_LIB=<path>).
    python scripts/build_variant.py <name> "<-Dflags>" <tu.hip> [<tu.hip> ...]"""
import importlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("online-3d-bpp-pct_amd.build")
name, defs, tus = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
vdir = os.path.join(b.HERE, "build", "v")
os.makedirs(vdir, exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-function-calls=false",
         "-Wno-pass-failed"]


def one(tu):
    obj = os.path.join(vdir, "%s_%s.o" % (name, os.path.splitext(tu)[0]))
    subprocess.check_call([b._hipcc(), *flags, *defs, "-c", os.path.join(b.CSRC, tu), "-o", obj])
    return obj


with ThreadPoolExecutor(max_workers=max(1, len(tus))) as ex:
    objs = dict(zip(tus, ex.map(one, tus)))
link = [objs.get(s, os.path.join(b.HERE, "build", os.path.splitext(s)[0] + ".o")) for s in b.SOURCES]
out = os.path.join(vdir, "lib%s.so" % name)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *link, "-o", out])
print(out)
