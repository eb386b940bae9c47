import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ppo_cpp_amd
E, T = int(os.environ.get("E", "4096")), 16
H = [int(x) for x in os.environ.get("HIDDEN", "256,256").split(",")]
g = ppo_cpp_amd.PPOHip(18, 18, H); g.init_orthogonal(0); g.norm_init(E); g.rollout_alloc(E, T)
rng = np.random.RandomState(0)
obs = rng.uniform(-1, 1, (E, 18)).astype(np.float32); rew = rng.uniform(-1, 1, E).astype(np.float32); dn = np.zeros(E, np.float32)
g.rollout_reset(obs)
for gap_us in (0, 150, 1000):
    ta = to = 0.0
    for rep in range(5):
        for t in range(T):
            t0 = time.perf_counter(); g.rollout_act(t); t1 = time.perf_counter(); g.rollout_observe(t, obs, rew, dn); t2 = time.perf_counter()
            if rep: ta += t1 - t0; to += t2 - t1
            if gap_us:
                e = time.perf_counter() + gap_us * 1e-6
                while time.perf_counter() < e: pass
    print("E=%d host gap %4d us: rollout_act %.1f us  rollout_observe %.1f us" % (E, gap_us, 1e6 * ta / (4 * T), 1e6 * to / (4 * T)))
g.prof_enable(True)
for t in range(T):
    g.rollout_act(t); g.rollout_observe(t, obs, rew, dn)
g.rollout_finish(0.99, 0.95)
print({k: (round(1e3 * ms / n, 1), n) for k, (ms, n) in g.prof_read().items() if n})
