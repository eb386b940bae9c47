import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ppo_cpp_amd, time
g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0)
g.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
g.norm_init(4096); g.rollout_alloc(4096, 16)
g.collect_synthetic(1234, 0.99, 0.95)
for i in range(3):
    t=time.perf_counter(); r,m = g.update(3e-4, 0.16, 10, 32, None, seed=i, want_rows=False); print("update ms %.2f" % (1e3*(time.perf_counter()-t)), m)
