#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, updates at the reference's own shape (1 environment x 2048 steps, [64,64], 32 minibatches of 64 rows), per-phase cycles of the
second-to-last minibatch step inside narrow_epoch_kernel (ppo_narrow.hpp).  usage: python tools/stamps_narrow_epoch.py [O]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS", "-o", so,
                       os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split(), stderr=subprocess.DEVNULL)
import ppo_cpp_amd
O = int(sys.argv[1]) if len(sys.argv) > 1 else 18
E, T, nmb = (int(sys.argv[2]), int(sys.argv[3]), 32) if len(sys.argv) > 3 else (1, 2048, 32)       # (1024 64: configs[3], narrow_epoch_dist_kernel)
g = ppo_cpp_amd.PPOHip(O, 18, [64, 64]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=0, first=True)
rng = np.random.RandomState(0); o_ = rng.uniform(-1, 1, (64, O)).astype(np.float32); a_, v_, n_ = g.step(o_, rng.normal(size=(64, 18)).astype(np.float32))
g.train_step(3e-4, 0.16, o_, a_, v_ * 0, v_, n_, v_)            # (allocates the stamp buffer outside the graph capture)
for i in range(2): g.update(3e-4, 0.16, 2, nmb, None, seed=i, want_rows=False)
dist = E * T // nmb > 64
buf = np.zeros(128 * 32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.reshape(128, 32).astype(np.int64)
if dist:
    G = (E * T // nmb + 31) // 32
    names = [(0, "step entry"), (11, "matrix phases (forward .. vectors)"), (16, "-"), (17, "drain + word + meeting 1"), (23, "rows to LDS + this workgroup's chunks"), (18, "drain + word + meeting 2"),
             (24, "gradient + sums of squares loaded"), (19, "norm + Adam + LDS image")]
    for tw in (0, 1):
        blk = st[tw * G:(tw + 1) * G]
        print("tower %d: step median %d cycles" % (tw, np.median(blk[:, 19] - blk[:, 0])))
        prev = 0
        for i, nm in names[1:]:
            dd = blk[:, i] - blk[:, prev]
            print("   %-44s median %7d  min %7d  max %7d (group %d)" % (nm, np.median(dd), dd.min(), dd.max(), int(dd.argmax())))
            if i == 23: print("      slowest:", ", ".join("g%d %d" % (j, dd[j]) for j in np.argsort(-dd)[:8]))
            prev = i
    g.close(); sys.exit(0)
st = st[:32]
st = st[::8] if st[8, 0] else st[:4]            # the XCD-local form's workgroups are the launch's workgroups 0, 8, 16, 24
order = [(0, "step entry"), (1, "barrier"), (2, "forward L0"), (3, "forward L1"), (6, "policy head"), (7, "policy loss"), (8, "head backward"), (9, "hidden backward"), (10, "dW"), (11, "vectors"),
         (16, "drain + barrier"), (17, "word + meeting"), (23, "partial loads + next rows staged"), (18, "assembly: sums, chunk trees"), (19, "norm + Adam + LDS image"), (22, "end of step")]
for w in range(4):
    print("workgroup %d (tower %d, group %d): step %d cycles" % (w, w // 2, w % 2, st[w, 22] - st[w, 0]))
    prev = 0
    for i, nm in order[1:]:
        if st[w, i] == 0: continue
        print("   %-44s %7d" % (nm, st[w, i] - st[w, prev])); prev = i
g.close()
