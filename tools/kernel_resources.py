#!/usr/bin/env python3
"""Register / LDS / scratch usage per gfx950 kernel (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
   python tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "diffusestylegesture_amd", "csrc", "dsg_hip.cpp")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only", "-c", src,
                      "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
names = re.findall(r"Function Name: (\S+)", out)
f = lambda key: [int(x) for x in re.findall(key + r": (\d+)", out)]
vg, ag, sc, lds, occ = f(r"  VGPRs"), f(r"AGPRs"), f(r"ScratchSize \[bytes/lane\]"), f(r"LDS Size \[bytes/block\]"), f(r"Occupancy \[waves/SIMD\]")
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
pats = sys.argv[1:]
print(f"{'VGPR':>5} {'AGPR':>5} {'scr':>4} {'LDS':>7} {'occ':>3}  kernel")
for i, n in enumerate(names):
    d = dem[i].replace("dsg::", "").replace("void ", "")
    d = re.sub(r"\(.*\)$", "", d)
    if pats and not any(p in d for p in pats):
        continue
    print(f"{vg[i]:5d} {ag[i]:5d} {sc[i]:4d} {lds[i]:7d} {occ[i]:3d}  {d}")
