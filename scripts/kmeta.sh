#!/bin/bash
# compile-only probe: VGPRs / spills / scratch of the benchmark's stability kernels under extra -D flags
#   scripts/kmeta.sh <tag> "<flags>" [pct_discrete_stab.hip|pct_continuous.hip]
TAG=$1; FLAGS=$2; TU=${3:-pct_discrete_stab.hip}
cd /root/repo/online-3d-bpp-pct_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-function-calls=false -Wno-pass-failed -DPCT_FEW_KERNELS $FLAGS -S --cuda-device-only $TU -o /tmp/kmeta_$TAG.s 2>/tmp/kmeta_$TAG.err || { tail -5 /tmp/kmeta_$TAG.err; exit 1; }
python3 - "$TAG" <<'PY'
import re, sys
tag = sys.argv[1]
txt = open("/tmp/kmeta_%s.s" % tag).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    name = m.group(1)
    k = re.search(r"kernelI[a-zA-Z]*Li\d+ELi(\d+)|continuous_kernelILi(\d+)", name)
    print("%-8s %s scratch %5s B  sgpr_spill %4s  vgpr %4s  vgpr_spill %4s" % (tag, name[:64], m.group(2), m.group(3), m.group(4), m.group(5)))
PY
grep -c "v_accvgpr" /tmp/kmeta_$TAG.s | sed "s/^/$TAG accvgpr instrs: /"
