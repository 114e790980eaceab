#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels in one HIP source, as the compiler reports them (-Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py coala-gnn_amd/csrc/coala_cache.hip [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
want = sys.argv[2:] or [""]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(ROOT, "include"),
       "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("EXTRA", "").split() + [src, "-o", "/dev/null"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
VG, AG, SG, SC, OC = "VGPRs", "AGPRs", "SGPRs", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]"
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    d2 = re.sub(r"\(anonymous namespace\)::", "", d)
    d2 = re.sub(r"^void ", "", re.sub(r"\(.*", "", d2))
    if any(w in d2 for w in want):
        print(f"{d2:100s} VGPR {g(VG):>3s} AGPR {g(AG):>3s} SGPR {g(SG):>3s} scratch {g(SC):>4s} occupancy {g(OC)}")
