#!/usr/bin/env python3
"""Diagnostic: which train8_kernel workgroups are the slow ones?  -DPPO_STAMPS build, train steps at the cfg3 shape; per-workgroup life (stamp 10 - stamp 0) by XCD
(lid & 7) and position in the XCD (lid >> 3), and the phases of the slowest workgroups against the median."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() +
                      ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
n = 2048; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
life = []
for rep in range(6):
    for _ in range(3): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
    buf = np.zeros(256 * 32, np.uint64)
    g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
    st = buf.reshape(256, 32).astype(np.int64)           # row = tower * 128 + rb
    life.append(st)
st = life[-1]
tot = st[:, 10] - st[:, 0]
lid = np.zeros(256, int)
for l in range(256):
    x, q = l & 7, l >> 3
    lid[(x & 1) * 128 + (x >> 1) * 32 + q] = l
print("train8_kernel workgroup life, cycles (last of 6 repetitions); rows of the stamp table mapped back to the launch's linear workgroup index")
for tower in (0, 1):
    t = tot[tower * 128:(tower + 1) * 128]
    print(" tower %d: median %d  p90 %d  max %d" % (tower, np.median(t), np.percentile(t, 90), t.max()))
    for x in range(tower, 8, 2):
        rows = [r for r in range(tower * 128, tower * 128 + 128) if (lid[r] & 7) == x]
        tt = tot[rows]
        print("   XCD %d: median %d max %d ; slowest positions (lid >> 3): %s" % (x, np.median(tt), tt.max(), [(int(lid[rows[i]] >> 3), int(tt[i])) for i in np.argsort(-tt)[:4]]))
# is the slow set stable across repetitions?
slow = [set(np.argsort(-(s[:, 10] - s[:, 0]))[:16]) for s in life]
print(" overlap of the 16 slowest workgroups between consecutive repetitions:", [len(slow[i] & slow[i + 1]) for i in range(len(slow) - 1)])
names = ["prologue", "fwd L0", "fwd L1", "head..", "", "", "loss", "head bwd", "bwd L1", "bwd L0 / db"]
t0 = tot[:128]
sl = np.argsort(-t0)[:8]
med = np.median(st[:128, 1:11] - st[:128, 0:10], axis=0)
print(" policy tower, phase durations: median | mean of the 8 slowest workgroups")
for i in range(10):
    d = (st[:128, i + 1] - st[:128, i])
    if (st[:128, i + 1] > 0).all() and (st[:128, i] > 0).all():
        print("   stamp %2d -> %2d  %7.0f | %7.0f" % (i, i + 1, np.median(d), d[sl].mean()))
