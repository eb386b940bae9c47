#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, policy steps of 4096 rows at cfg3's shape, per-phase cycles of policy_step_kernel (last launch)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] +
                      os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
n = int(os.environ.get("ROWS", 4096))
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); noise = rng.normal(size=(n, 18)).astype(np.float32)
for _ in range(4): g.step(obs, noise)
nb = n // 16
buf = np.zeros(2 * nb * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -(3 << 20) - buf.size)
st = buf.reshape(2, nb, 8).astype(np.int64)
for tower, nm in ((0, "policy tower"), (1, "value tower")):
    b = st[tower]
    print(" %s: %d workgroups, whole workgroup median %d max %d cycles" % (nm, len(b), np.median(b[:, 5] - b[:, 0]), (b[:, 5] - b[:, 0]).max()))
    names = ["prologue (kernel arguments, inputs, first weights)", "first layer", "second layer", "policy head" if tower == 0 else "value head + store", "sample, neglogp, stores"]
    for i, name in enumerate(names):
        if tower == 1 and i == 3: d = b[:, 5] - b[:, 3]
        elif tower == 1 and i == 4: continue
        else: d = b[:, i + 1] - b[:, i]
        print("   %-52s median %6d max %6d" % (name, np.median(d), d.max()))
