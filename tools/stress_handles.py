import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, ppo_cpp_amd, time
import ctypes
def rss():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e6
t0 = time.time()
for i in range(40):
    g = ppo_cpp_amd.PPOHip(18, 18, [64, 64] if i % 2 else [256, 256])
    g.init_orthogonal(i); g.norm_init(64); g.rollout_alloc(64, 16)
    g.collect_synthetic(1, 0.99, 0.95)
    rows, mean = g.update(3e-4, 0.16, 2, 4, None, seed=i)
    assert np.isfinite(rows).all()
    g.close()
    if i % 10 == 9: print(i, "rss MB %.0f" % rss(), "t %.1f" % (time.time() - t0), flush=True)
hip = ctypes.CDLL("libamdhip64.so")
free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
print("device free MB", free.value / 1e6, "of", total.value / 1e6)
