#!/usr/bin/env python3
"""Register and scratch use of every kernel in rt_kernels.hip, from the compiler's own remarks (`make -C raytracing_amd/csrc asm`)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(root, "raytracing_amd", "csrc"), "asm"], capture_output=True, text=True)
cur = None
rows = {}
for line in (out.stdout + out.stderr).splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?):\s+(\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-62s %5s %5s %7s %7s %8s %4s" % ("kernel", "VGPR", "SGPR", "spill V", "spill S", "scratch", "occ"))
for k, r in rows.items():
    print("%-62s %5d %5d %7d %7d %8d %4d" % (k.replace("rt::", ""), r.get("VGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                            r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
