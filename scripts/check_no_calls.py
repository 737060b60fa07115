#!/usr/bin/env python
"""Build check: no kernel of libpct_hip.so may contain a real function call.

profiles/r03_fault_root_cause.txt: this hipcc places AGPR split copies around a call on the wrong side of the exec
restore, so a kernel that calls (s_swappc_b64) a non-inlined device function under register pressure can lose live
registers.  The library is built with -mllvm -amdgpu-function-calls=false; this script disassembles the gfx950 code
object inside the built library and fails if any s_swappc_b64 / s_call_b64 is left (s_setpc_b64 alone is how long branches are relaxed).  It also prints the
register / scratch / LDS footprint of the kernels whose names match the optional filter.
    python scripts/check_no_calls.py [library] [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _llvm_bin():
    """llvm-objdump & friends next to the hipcc the build uses ($HIPCC, PATH, $ROCM_PATH, /opt/rocm)"""
    import shutil
    cands = []
    for hipcc in (os.environ.get("HIPCC"), shutil.which("hipcc")):
        if hipcc:
            cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin"), "/opt/rocm/lib/llvm/bin"]
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    return cands[-1]


LLVM = _llvm_bin()


def device_code_objects(lib, tmp):
    """every gfx950 code object in the library: one clang offload bundle per translation unit, concatenated in .hip_fatbin"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, "bundle%d.bin" % i)
        open(part, "wb").write(blob[a:b])
        co = os.path.join(tmp, "dev%d.co" % i)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co])
        out.append(co)
    return out


def kernel_resources(lib):
    """[(mangled name, vgprs, sgprs, scratch bytes, spilled vgprs)] of every kernel in the library"""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in device_code_objects(lib, tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                def g(key):
                    m = re.search(r"\." + key + r":\s+(\S+)", k)
                    return m.group(1) if m else None
                rows.append((g("name"), int(g("vgpr_count") or 0), int(g("sgpr_count") or 0),
                             int(g("private_segment_fixed_size") or 0), int(g("vgpr_spill_count") or 0)))
    return rows


def count_calls(lib):
    calls, kernels = {}, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in device_code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            name = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    name = m.group(1)
                    if name.startswith("_Z") and "kernel" in name:
                        kernels += 1
                    continue
                if re.search(r"\bs_(swappc|call)_b64\b", ln):  # (s_setpc_b64 alone is a long branch, not a call)
                    calls[name] = calls.get(name, 0) + 1
    return calls, kernels


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "online-3d-bpp-pct_amd", "libpct_hip.so")
    calls, kernels = count_calls(lib)
    if calls:
        for n, c in sorted(calls.items()):
            print("CALL x%d in %s" % (c, n))
        print("FAILED: %d functions of %s contain real calls" % (len(calls), lib))
        return 1
    print("ok: %d kernels in %s, no s_swappc_b64 / s_call_b64" % (kernels, os.path.basename(lib)))
    if len(sys.argv) > 2:
        for name, v, sg, scr, sp in sorted(kernel_resources(lib)):
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if sys.argv[2] in dn:
                print("%-100s vgpr %3d sgpr %3d scratch %5d spilled %3d" % (dn[:100], v, sg, scr, sp))
    return 0


if __name__ == "__main__":
    sys.exit(main())
