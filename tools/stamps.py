#!/usr/bin/env python3
"""Diagnostic: builds libppo_hip with -DPPO_STAMPS, runs train steps at cfg3 shape, prints per-phase cycles."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "libppo_hip_stamps.so")            # never over the product library (bench.py / pytest keep loading the real one)
os.environ["PPO_HIP_LIBRARY"] = so
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DPPO_STAMPS", "-DPPO_STAMP_LAYER=" + os.environ.get("STAMP_LAYER", "1")] + os.environ.get("PPO_HIP_EXTRA_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "ppo_cpp_amd/csrc/ppo_hip.hip"), "-ldl"])
import ppo_cpp_amd
H = [int(x) for x in os.environ.get("HIDDEN", "256,256").split(",")]
g = ppo_cpp_amd.PPOHip(18, 18, H); g.init_orthogonal(0)
n = 2048; rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (n, 18)).astype(np.float32); a, v, nlp = g.step(obs, rng.normal(size=(n, 18)).astype(np.float32))
ret = (v + rng.normal(size=n)).astype(np.float32); adv = g.adv_normalize(ret, v)
for _ in range(5): g.train_step(3e-4, 0.16, obs, a, adv, ret, nlp, v)
buf = np.zeros(256 * 32, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
st = buf.reshape(256, 32).astype(np.int64)
labels = [(0, 1, "prologue (stage)"), (1, 2, "fwd L0"), (2, 3, "fwd L1"), (3, 6, "pi head fwd"), (6, 7, "pi loss"), (3, 7, "vf head+loss"), (7, 8, "head bwd"),
          (8, 9, "bwd L1 (dX)"), (9, 10, "bwd L0 (db)")]
for tower in (0, 1):
    blk = st[tower * 128:(tower + 1) * 128]
    print("tower", tower, "kernel cycles median", np.median(blk[:, 10] - blk[:, 0]), "max", (blk[:, 10] - blk[:, 0]).max())
    for i, j, name in labels:
        if (blk[:, i] > 0).all() and (blk[:, j] > 0).all():
            d = blk[:, j] - blk[:, i]
            print("   %-18s median %7.0f  max %7.0f" % (name, np.median(d), d.max()))
    for i, j, name in [(0, 20, "  kernarg round trip"), (20, 21, "  issue loads"), (21, 22, "  wait + consume"), (22, 1, "  barrier"),
                       (2, 11, "  fwd L1 product"), (11, 12, "  issue W1T stages"), (12, 13, "  exchange"), (13, 3, "  bias+tanh+stores+barrier"),
                       (8, 14, "  bwd L1 product"), (14, 15, "  exchange"), (15, 9, "  TanhGrad+stores+barrier")]:
        if (blk[:, i] > 0).all() and (blk[:, j] > 0).all():
            d = blk[:, j] - blk[:, i]
            print("   %-28s median %7.0f  max %7.0f" % (name, np.median(d), d.max()))
    if os.environ.get("STAMPS_T8_ONLY"): continue
    seq = [16, 20, 21, 22, 23, 24, 25, 26, 27, 17, 18, 19]
    names = ["L1 entry", "st0", "st1", "st2", "st3", "st4", "st5", "st6", "st7", "loop end", "between done", "epilogue done"]
    prev = None
    for idx, nm in zip(seq, names):
        if (blk[:, idx] > 0).all():
            if prev is not None: print("      %-14s +%6.0f" % (nm, np.median(blk[:, idx] - blk[:, prev])))
            prev = idx
if os.environ.get('STAMPS_T8_ONLY'): sys.exit(0)
buf = np.zeros(256 * 8, np.uint64)
g.lib.ppo_debug_read_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), -buf.size)
sb = buf.reshape(-1, 8).astype(np.int64)
big = sb[(sb[:, 4] > 0)]
t0 = big[:, 0].min()
print("weight_grad 64x64 tiles: %d blocks; start spread %d, last end %d cycles after first start" % (len(big), (big[:, 0] - t0).max(), (big[:, 4] - t0).max()))
for i, nm in enumerate(["issue prologue loads", "main loop", "lds park + barrier", "sum + slab store"]):
    d = big[:, i + 1] - big[:, i]
    print("   %-22s median %7.0f max %7.0f" % (nm, np.median(d), d.max()))

print("kernel B whole-kernel: entry->end span %.2f us ; entry->body-start median %.2f us ; body median %.2f us ; after-body (strips) median %.2f us ; entry skew %.2f us" % (
    (big[:, 6].max() - big[:, 5].min()) / 100.0, 0.0, np.median(big[:, 7] - big[:, 5]) / 100.0, np.median(big[:, 6] - big[:, 7]) / 100.0,
    (big[:, 5].max() - big[:, 5].min()) / 100.0))
