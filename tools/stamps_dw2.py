#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, train steps at the cfg3 shape, per-phase cycles of weight_grad_assemble_kernel."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() +
                      ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
n = 2048; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(5): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
buf = np.zeros(256 * 16, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -buf.size)
st = buf.reshape(256, 16).astype(np.int64)
names = ["entry -> chunk 0 requested", "chunk 0 landed (+ slot loads issued)", "chunk loop", "park + sums + slab stores issued", "drain + barrier", "arrival", "last arriver"]
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print("   %-34s median %7.0f  p90 %7.0f  max %7.0f" % (nm, np.median(d), np.percentile(d, 90), d.max()))
tot = st[:, 7] - st[:, 0]
print("   whole workgroup: median %d max %d cycles ; last arrivers (phase 6 > 500 cycles): %d" % (np.median(tot), tot.max(), ((st[:, 7] - st[:, 6]) > 500).sum()))
rt = st[:, 15]
print("   entry skew (s_memrealtime, 10 ns ticks): %d" % (rt.max() - rt.min()))
if st[:, 8].any():            # clip + Adam inside the launch (Dw2Adam): stamps 8 (tile work done), 9 (meeting over), 10 (norm known), 7 (applied)
    fin = (st[:, 8] - st[:, 6]) > 500
    for nm, a, b in (("tile finisher's work (finishers)", 6, 8), ("drain + word + meeting", 8, 9), ("partials -> norm", 9, 10), ("apply + stores issued", 10, 7)):
        d = st[:, b] - st[:, a]
        d = d[fin]                # (only the finishers go on past stamp 8)
        print("   %-34s median %7.0f  p90 %7.0f  max %7.0f  min %7.0f" % (nm, np.median(d), np.percentile(d, 90), d.max(), d.min()))
    k = np.array([0 if (b >> 3) < 16 else (1 if (b & 1) == 0 else 2) for b in range(256)])
    d = (st[:, 7] - st[:, 10])
    print("   apply by strip kind (first layer / head / none): " + " / ".join("%d" % np.median(d[fin & (k == q)]) for q in range(3)))
