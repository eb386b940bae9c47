#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, one ppo_update at [64,64] (the deferred-Adam train kernel), per-phase cycles of the LAST
train launch, then rebuilds the production library."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
subprocess.check_call(base + ["-DPPO_STAMPS", "-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split())
import ppo_cpp_amd
E, T, nmb = (int(sys.argv[1]), int(sys.argv[2]), 32) if len(sys.argv) > 2 else (1024, 64, 32)
g = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=0, first=True)
rng = np.random.RandomState(0); o_ = rng.uniform(-1, 1, (64, 18)).astype(np.float32); a_, v_, n_ = g.step(o_, rng.normal(size=(64, 18)).astype(np.float32))
g.train_step(3e-4, 0.16, o_, a_, v_ * 0, v_, n_, v_)            # (allocates the stamp buffer outside the graph capture)
for i in range(2): g.update(3e-4, 0.16, 2, nmb, None, seed=i, want_rows=False)
G = (E * T // nmb + 31) // 32
buf = np.zeros(2 * G * 32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.reshape(2 * G, 32).astype(np.int64)
order = [(0, "entry"), (12, "lazy: loads issued"), (13, "rows staged (first wait)"), (20, "norm barrier + scale"), (21, "Adam math + LDS image + write-back"), (14, "block-0 tail"), (1, "barrier"),
         (2, "forward L0"), (3, "forward L1"), (6, "policy head"), (7, "policy loss"), (8, "head backward"), (9, "hidden backward"), (10, "dW"), (11, "vectors + end")]
for tower in (0, 1):
    blk = st[tower * G:(tower + 1) * G]
    print("tower", tower, "kernel cycles median", np.median(blk[:, 11] - blk[:, 0]), "max", (blk[:, 11] - blk[:, 0]).max())
    prev = 0
    for i, nm in order[1:]:
        if not (blk[:, i] > 0).all(): continue
        d = blk[:, i] - blk[:, prev]
        print("   %-44s median %7.0f  max %7.0f" % (nm, np.median(d), d.max()))
        prev = i
g.close()
subprocess.check_call(base + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
