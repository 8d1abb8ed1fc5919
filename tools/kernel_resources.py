#!/usr/bin/env python3
"""Prints VGPR / spill / LDS / occupancy of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "isaac_ros_apriltag_amd", "csrc", "detector.hip"), "-o", "/tmp/_res.so"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+\w)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-36s %6s %6s %8s %8s %6s %8s" % ("kernel", "VGPR", "AGPR", "spillV", "scratch", "occ", "LDS"))
for k, r in rows.items():
    print("%-36s %6d %6d %8d %8d %6d %8d" % (k[:36], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("VGPRs Spill", r.get("VGPR Spill", 0)),
                                          r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
if "error" in err:
    print(err[-3000:])
