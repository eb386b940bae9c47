#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, train steps at cfg3's shape, per-phase cycles of adam_kernel (last launch)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] +
                      os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
n = 2048; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(5): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
nb = 160
buf = np.zeros(nb * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -(2 << 20) - buf.size)
st = buf.reshape(nb, 8).astype(np.int64)
st = st[st[:, 0] > 0]
print("%d workgroups" % len(st))
for i, nm in enumerate(["tile lookup + element loads issued", "hyper-parameters + partial sums of squares loaded, summed", "wave sums -> LDS, barrier", "norm, clip, Adam arithmetic, stores issued",
                        "transposed tile through LDS", "stores drained"]):
    d = st[:, i + 1] - st[:, i]
    print("   %-58s median %6d max %6d" % (nm, np.median(d), d.max()))
print("   whole workgroup median %d max %d cycles" % (np.median(st[:, 6] - st[:, 0]), (st[:, 6] - st[:, 0]).max()))
