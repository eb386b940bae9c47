#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, train steps at the cfg3 shape through train8_dw2_fused_kernel; cycles of phase A (train8 body), the grid-wide
meeting and phase B (weight-gradient body) per workgroup.  PPO_HIP_FUSE_AB=0: the same stamps from the two separate launches."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
os.environ.setdefault("PPO_HIP_FUSE_AB", "1")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() +
                      ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
n = 2048; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(5): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
fused = g.kernel_counts().get("train8_dw2_fused_kernel", 0) > 0
gx = 256 if fused else 128                                    # STAMP's row = tower * gridDim.x + rb
buf = np.zeros((gx + 128) * 32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
sa = buf.reshape(-1, 32).astype(np.int64)
buf = np.zeros(256 * 16, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -buf.size)
sb = buf.reshape(256, 16).astype(np.int64)
A0 = np.zeros(256, np.int64); A1 = np.zeros(256, np.int64)
for lid in range(256):
    x, q = lid & 7, lid >> 3
    tower, rb = x & 1, (x >> 1) * 32 + q
    A0[lid], A1[lid] = sa[tower * gx + rb, 0], sa[tower * gx + rb, 10]
B0, B1, Bl, Bp = sb[:, 0], sb[:, 7], sb[:, 3], sb[:, 1]
print("fused launch" if fused else "two launches")
for tower in (0, 1):
    m = (np.arange(256) & 1) == tower
    print("  tower %d: phase A %6.0f median %6.0f max cycles" % (tower, np.median((A1 - A0)[m]), (A1 - A0)[m].max()))
print("  phase B: whole %6.0f median %6.0f max | entry -> chunk 0 landed %5.0f | chunk loop %6.0f | tail %5.0f" % (
    np.median(B1 - B0), (B1 - B0).max(), np.median(sb[:, 2] - B0), np.median(sb[:, 3] - sb[:, 2]), np.median(B1 - sb[:, 3])))
if fused:
    w = B0 - A1
    print("  meeting (end of phase A -> start of phase B): median %5.0f min %5.0f max %5.0f ; last arrival -> first release %5.0f ; release spread %5.0f" % (
        np.median(w), w.min(), w.max(), B0.min() - A1.max(), B0.max() - B0.min()))
    print("  whole launch (first entry -> last exit): %6.0f cycles" % (B1.max() - A0.min()))
