#!/usr/bin/env python
"""Prints the resource metadata (VGPRs, SGPRs, scratch, LDS, spills) of every kernel in a gfx950 assembly file
(hipcc --offload-device-only -S).  Used to keep an eye on register pressure / scratch of the kernel variants."""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n


def main(path):
    txt = open(path).read()
    md = txt[txt.rfind(".amdgpu_metadata"):]
    for k in re.split(r"\n  - \.agpr_count", md)[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", k)
            return m.group(1) if m else None
        print("%-110s vgpr %s sgpr %s scratch %s lds %s vspill %s sspill %s dynstack %s" % (
            demangle(name)[:110], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"),
            g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("uses_dynamic_stack")))


if __name__ == "__main__":
    main(sys.argv[1])
