#!/usr/bin/env python3
"""VGPR / scratch / occupancy / LDS of every kernel of a HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).  python tools/kernel_resources.py blockmaze_amd/csrc/gpu_msm_g1.hip"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1]); out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
cur = None; d = {}
for l in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", l)
    if m: cur = m.group(1); d[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*): (\w+)", l)
    if m and cur: d[cur][m.group(1).strip()] = m.group(2)
for k, v in d.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("zk::", "").replace("Fp<FqParams>", "Fq").replace("Fp<FrParams>", "Fr")[:64]
    print("%-66s VGPR %3s AGPR %3s scratch %5s occupancy %s LDS %6s" % (name, v.get("VGPRs"), v.get("AGPRs"), v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
