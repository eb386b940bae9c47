#!/usr/bin/env python3
"""Diagnostic: -DPPO_STAMPS build, rollouts at BASELINE configs[4]'s observation shape (8192 x 256, bf16 handle), per-phase cycles of norm_batch_kernel's
column-group observation job (obs_cgroup_job) in the last launch."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPPO_STAMPS"] +
                      os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
E, T, O = int(os.environ.get("E", 8192)), 4, int(os.environ.get("O", 256))
g = ppo_cpp_amd.PPOHip(O, 64, [1024, 1024, 1024], compute_dtype=1); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
for i in range(3): g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=i * T, first=(i == 0))
nb = 1100
buf = np.zeros(nb * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -(1 << 20) - buf.size)
st = buf.reshape(nb, 8).astype(np.int64)
live = st[(st[:, 0] > 0) & (st[:, 1] > 0)]
print("%d workgroups with strip stamps (cycle counters are per XCD: only differences inside a workgroup are compared)" % len(live))
names = ["loads + two passes", "set stores + drain + barrier", "group arrival", "group-last: combine + merge", "drain + barrier", "global arrival"]
ok = live[(live[:, 7] > 0) & (live[:, 4] > 0)]
if len(ok): print("   group-last: combine %6d, merge / publish %6d (medians)" % (np.median(ok[:, 7] - ok[:, 3]), np.median(ok[:, 4] - ok[:, 7])))
for i, nm in enumerate(names):
    ok = live[(live[:, i + 1] > 0) & (live[:, i] > 0) & (live[:, i + 1] >= live[:, i])]
    if len(ok):
        d = ok[:, i + 1] - ok[:, i]
        print("   %-32s n %4d median %6d max %6d" % (nm, len(ok), np.median(d), d.max()))
