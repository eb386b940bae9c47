#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, train steps at the cfg5 shape (256/64/[1024]^3, 4096 rows), per-phase cycles of gemm_nt_bf16_kernel."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() +
                      ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
g = ppo_cpp_amd.PPOHip(256, 64, [1024, 1024, 1024], compute_dtype=1); g.init_orthogonal(0)
n = 4096; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 256)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 64)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(4): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
buf = np.zeros(3 * 2 * 256 * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.reshape(3, 2, 256, 8).astype(np.int64)
names = ["entry -> first two stages requested", "k loop", "barrier after the loop", "operand loads + math + park", "barrier", "row reads + stores issued", "stores drained"]
for epi, nm in ((0, "TANH (last forward layer)"), (1, "TANHGRAD (last backward layer)")):
    s = st[epi].reshape(-1, 8); s = s[s[:, 0] > 0]
    print("%s: %d workgroups" % (nm, len(s)))
    for i, pn in enumerate(names):
        d = s[:, i + 1] - s[:, i]
        print("   %-38s median %7.0f  p90 %7.0f  max %7.0f" % (pn, np.median(d), np.percentile(d, 90), d.max()))
    tot = s[:, 7] - s[:, 0]
    print("   whole workgroup: median %d max %d cycles" % (np.median(tot), tot.max()))

buf = np.zeros(256 * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -buf.size)
d = buf.reshape(256, 8).astype(np.int64)
two = d[:, 3] > 0
print("gemm_dw_bf16_kernel (work-balanced): %d workgroups, %d with two segments" % (len(d), two.sum()))
l1 = d[:, 1] - d[:, 0]; st1 = d[:, 2] - d[:, 1]
print("   segment 1: k loop median %d cycles for median %d stages = %.0f cycles per stage ; stores issued %d" % (np.median(l1), np.median(d[:, 5]), np.median(l1 / np.maximum(d[:, 5], 1)), np.median(st1)))
if two.any():
    l2 = d[two, 3] - d[two, 2]; st2 = d[two, 4] - d[two, 3]
    print("   segment 2: k loop median %d cycles for median %d stages = %.0f cycles per stage ; stores issued %d" % (np.median(l2), np.median(d[two, 6]), np.median(l2 / np.maximum(d[two, 6], 1)), np.median(st2)))
last = np.where(two, d[:, 4], d[:, 2])
tot = last - d[:, 0]
print("   whole workgroup: median %d  p90 %d  max %d cycles" % (np.median(tot), np.percentile(tot, 90), tot.max()))
