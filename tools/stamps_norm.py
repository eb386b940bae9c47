#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, one rollout at cfg3's shape, per-phase cycles of norm_batch_kernel (last launch)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] +
                      os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
E, T = int(os.environ.get("E", 4096)), 16
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
for i in range(3): g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=i * T, first=(i == 0))
nb = 64
buf = np.zeros(nb * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -(1 << 20) - buf.size)
st = buf.reshape(nb, 8).astype(np.int64)
live = st[st[:, 0] > 0]
print("%d workgroups (cycle counters are per XCD: only differences inside a workgroup are compared)" % len(live))
for which, name in ((0, "observation job"), (1, "reward job")):
    blk = live[live[:, 6] == which]
    print(" %s: %d workgroups" % (name, len(blk)))
    for i, nm in enumerate(["chunk moments / return update", "release fence + barrier", "arrival"]):
        d = blk[:, i + 1] - blk[:, i]
        print("   %-32s median %6d max %6d" % (nm, np.median(d), d.max()))
    for row in blk[blk[:, 5] > 0]:
        print("   last arriver: acquire fence %5d, combine + merge%s %6d; its whole life %6d cycles" % (row[4] - row[3], " + reward apply" if which else "", row[5] - row[4], row[5] - row[0]))
