"""Soak / consistency check: N full collect+update iterations at config 3, graph replay vs eager launches, bitwise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_cpp_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
E, T = 4096, 16
def run(eager):
    if eager: os.environ["PPO_HIP_NO_GRAPH"] = "1"
    else: os.environ.pop("PPO_HIP_NO_GRAPH", None)
    g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
    t0 = time.time(); means = []
    for i in range(N):
        g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=i * T, first=(i == 0))
        means.append(g.update(3.93141e-4, 0.161023, 10, 32, None, seed=1000 + i, want_rows=False)[1].copy())
    th = g.get_flat(0); st = g.norm_stats(0); dt = time.time() - t0
    g.close()
    return th, np.array(means), st, dt
a = run(False); b = run(True)
print("graph %.2f s, eager %.2f s for %d iterations (%.3g env-steps/s graph)" % (a[3], b[3], N, N * E * T / a[3]))
print("finite:", np.isfinite(a[0]).all(), "bitwise equal weights:", np.array_equal(a[0], b[0]), "loss means equal:", np.array_equal(a[1], b[1]),
      "stats equal:", np.array_equal(a[2][0], b[2][0]) and a[2][2] == b[2][2])
print("loss means first/last:", a[1][0], a[1][-1])
assert np.isfinite(a[0]).all() and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
