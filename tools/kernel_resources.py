#!/usr/bin/env python3
"""Register / scratch / occupancy of every kernel in a .hip file (compile-time remark parse).
usage: kernel_resources.py coarse3d_amd/csrc/conv_mfma.hip"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
       "-fno-fast-math", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, rec = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").split("(")[0]
        rec[name] = {}
    for k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m and name:
            rec[name][k.split()[0]] = int(m.group(1))
for n, r in rec.items():
    print(f"{n[:72]:72s} vgpr {r.get('VGPRs'):4d} agpr {r.get('AGPRs'):4d} scratch {r.get('ScratchSize'):4d} occ {r.get('Occupancy')}")
