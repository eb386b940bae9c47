"""Leak check: handles of several shapes (fp32 fast pair, narrow, bf16; host layer) created, used for a rollout + update and destroyed N times;
free device memory and the process's resident set before and after."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ppo_cpp_amd
from ppo_cpp_amd import hostapi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
def free_mb(): return torch.cuda.mem_get_info(0)[0] / 2**20
def rss_mb(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
def once(i):
    for O, A, hidden, E, T, nmb, dt in ((18, 18, [256, 256], 1024, 8, 4, 0), (18, 18, [64, 64], 32, 16, 4, 0), (36, 18, [256, 256], 256, 8, 2, 0), (64, 16, [512, 512], 256, 8, 2, 1)):
        g = ppo_cpp_amd.PPOHip(O, A, hidden, compute_dtype=dt); g.init_orthogonal(i); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
        g.collect_synthetic(7, 0.99, 0.95, None, env0=0, step0=0, first=True)
        g.update(3e-4, 0.16, 2, nmb, None, seed=i, want_rows=False)
        g.close()
    hostapi.learn(64, 16, [64, 64], 2, nminibatches=4, noptepochs=2, seed=i)
once(0); once(1)
torch.cuda.synchronize(); f0, r0 = free_mb(), rss_mb(); t0 = time.time()
for i in range(N): once(i)
torch.cuda.synchronize(); f1, r1 = free_mb(), rss_mb()
print("%d rounds of 5 handles in %.1f s: free device memory %.1f -> %.1f MB (%+.1f), peak RSS %.1f -> %.1f MB (%+.1f)" % (N, time.time() - t0, f0, f1, f1 - f0, r0, r1, r1 - r0))
assert f0 - f1 < 64 and r1 - r0 < 256, "leak?"
