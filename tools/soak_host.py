"""Soak of the C++ host layer on the GPU: PPO2::learn over many updates behind the pooled VecEnv (large batch) and behind the resident
host-env kernel (one and a few environments); prints throughput and checks that the losses stay finite and that two runs give the same bits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ppo_cpp_amd import hostapi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for E, T, hidden, nmb in ((4096, 16, [256, 256], 32), (1, 256, [64, 64], 4), (20, 64, [64, 64], 4), (256, 32, [256, 256], 8)):
    runs = []
    for rep in range(2):
        t0 = time.time()
        r = hostapi.learn(E, T, hidden, N, nminibatches=nmb, seed=3)
        runs.append(r); dt = time.time() - t0
    a, b = runs
    same = all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in ("losses", "obs_count", "ret_count"))
    print("E %5d T %4d %s: %d updates in %.2f s (%.3g env-steps/s), losses finite %s, two runs identical %s, pool %s" % (
        E, T, hidden, N, dt, a["env_steps_per_s"], bool(np.isfinite(np.asarray(a["losses"])).all()), same, a.get("vec_env_pool")))
    assert np.isfinite(np.asarray(a["losses"])).all() and same
