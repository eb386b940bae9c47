#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, one update at cfg3's shape, per-phase cycles of epoch_prepare_gather_kernel (last launch)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] +
                      os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
E, T = 4096, 16
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=0, first=True)
g.update(3.9e-4, 0.16, 2, 32, None, seed=1, want_rows=False)
nb = 256
buf = np.zeros(nb * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -(1 << 20) - buf.size)
st = buf.reshape(nb, 8).astype(np.int64)
names = ["index map + first gather of returns / values", "sum -> mean", "squared deviations (second pass)", "sum -> denominator", "row gather (obs, actions)", "scalar fields", "stores drained"]
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print("   %-46s median %6d max %6d" % (nm, np.median(d), d.max()))
print("   whole workgroup median %d max %d cycles" % (np.median(st[:, 7] - st[:, 0]), (st[:, 7] - st[:, 0]).max()))
