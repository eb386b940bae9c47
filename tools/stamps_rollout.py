#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, one persistent rollout ([64,64], E envs), per-phase cycles of the LAST env step."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
subprocess.check_call(base + ["-DPPO_STAMPS", "-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split())
import ppo_cpp_amd
E, T = int(os.environ.get("ENVS", "1")), 256
g = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
for i in range(2): g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=i * T, first=(i == 0))
buf = np.zeros(32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.astype(np.int64)
names = ["normalise + stage", "forward L0", "forward L1", None, None, "head", "sample", "env transition", "statistics", "end barrier"]
prev = 0
for i, nm in enumerate(names, start=1):
    if nm is None or st[i] == 0: continue
    print("   %-24s %6d" % (nm, st[i] - st[prev])); prev = i
print("   step total %d cycles" % (st[10] - st[0]))
g.close()
subprocess.check_call(base + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
