#!/usr/bin/env python3
"""Diagnostic: builds libppo_hip with -DPPO_STAMPS, runs narrow-path train steps ([64,64], 2048 rows), prints per-phase cycles,
then rebuilds the production library."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
subprocess.check_call(base + ["-DPPO_STAMPS", "-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
H = [int(x) for x in os.environ.get("HIDDEN", "64,64").split(",")]
n = int(os.environ.get("ROWS", "2048"))
g = ppo_cpp_amd.PPOHip(18, 18, H); g.init_orthogonal(0)
rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(5): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
G = (n + 31) // 32
buf = np.zeros(2 * G * 32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.reshape(2 * G, 32).astype(np.int64)
L = len(H)
names = {0: "entry", 1: "stage + barrier", 6: "policy head", 7: "policy loss", 8: "head backward (vf: value head+loss+bwd)", 9: "hidden backward", 10: "dW matrices", 11: "vectors + end"}
for l in range(L): names[2 + l] = "forward L%d" % l
for tower in (0, 1):
    blk = st[tower * G:(tower + 1) * G]
    print("tower", tower, "kernel cycles median", np.median(blk[:, 11] - blk[:, 0]), "max", (blk[:, 11] - blk[:, 0]).max())
    prev = 0
    for i in sorted(names):
        if i == 0 or not (blk[:, i] > 0).all(): continue
        d = blk[:, i] - blk[:, prev]
        print("   %-44s median %7.0f  max %7.0f" % (names[i], np.median(d), d.max()))
        prev = i
g.close()
subprocess.check_call(base + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
