#!/usr/bin/env python
"""Compiler resource usage of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, AGPRs, scratch,
spills, occupancy.  usage: python tools/resource_usage.py hetmogp_amd/csrc/rowpass.hip [--all]   (default: only kernels with scratch / spills)"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
show_all = "--all" in sys.argv
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-c", src, "-o", "/tmp/_ru.o",
       "-Rpass-analysis=kernel-resource-usage"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
filt = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(p)), None)
rows = []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split(" [")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                 g(r"Occupancy \[waves/SIMD\]")))
for r in rows:
    d = subprocess.run([filt, r[0]], capture_output=True, text=True).stdout.strip() if filt else r[0]
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", d)
    if show_all or r[3] > 0 or r[4] > 0 or r[5] > 0:
        print("%-70s VGPR %3d AGPR %3d scratch %4d B  SGPR-spill %3d  VGPR-spill %3d  occupancy %d" % (d[:70], r[1], r[2], r[3], r[4], r[5], r[6]))
print("%d kernels in %s" % (len(rows), src))
