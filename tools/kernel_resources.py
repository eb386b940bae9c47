#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/scratch/occupancy per kernel of libppo_hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_res.o", os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (.*?) \[-R", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur: rows[cur][k.split(" [")[0]] = v
print("%-34s %5s %5s %5s %8s %6s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "spill", "occ"))
for k, r in rows.items():
    print("%-34s %5s %5s %5s %8s %6s %5s" % (k[-34:], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("Occupancy")))
