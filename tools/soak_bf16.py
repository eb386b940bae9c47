"""Soak / consistency check of the bf16 path: N collect + update iterations at 256 obs / 64 act / [1024]^3, hipGraph replay against eager
launches, bitwise (every kernel of the path sums in a fixed order: the work-balanced weight-gradient GEMM's slabs included).
usage: python tools/soak_bf16.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_cpp_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
def run(E, T, nmb, eager):
    if eager: os.environ["PPO_HIP_NO_GRAPH"] = "1"
    else: os.environ.pop("PPO_HIP_NO_GRAPH", None)
    g = ppo_cpp_amd.PPOHip(256, 64, [1024, 1024, 1024], compute_dtype=1); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
    means = []
    for i in range(N):
        g.collect_synthetic(1234, 0.99, 0.95, None, env0=0, step0=i * T, first=(i == 0))
        means.append(g.update(3.93141e-4, 0.161023, 2, nmb, None, seed=1000 + i, want_rows=False)[1].copy())
    th = g.get_flat(0); g.close()
    return th, np.array(means)
for E, T, nmb in ((2048, 16, 8), (512, 16, 4), (96, 8, 2)):          # 4096-, 2048- and 384-row minibatches (256- and 128-row tiles)
    t0 = time.time(); a = run(E, T, nmb, False); b = run(E, T, nmb, True)
    ok = np.isfinite(a[0]).all() and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print("E %4d T %3d: %d iterations graph vs eager bitwise equal: %s (%.1f s); loss means first / last %s / %s" % (E, T, N, ok, time.time() - t0, a[1][0], a[1][-1]), flush=True)
    assert ok
