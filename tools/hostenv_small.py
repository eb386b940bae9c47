"""Host-Env rollout, small environment counts: us per env step (free env: constant transitions) for the resident kernel, the
fused launch per step (<= 32 envs) and the general path.   usage: python tools/hostenv_small.py [E ...]"""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import ctypes as C, ppo_cpp_amd
    E, T = int(sys.argv[2]), 256
    g = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
    rng = np.random.RandomState(0)
    obs = rng.uniform(-1, 1, (E, 18)).astype(np.float32); rew = rng.uniform(-1, 1, E).astype(np.float32); dn = np.zeros(E, np.float32)
    act = np.zeros((E, 18), np.float32); fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    g._ck(g.lib.ppo_rollout_reset(g.h, fp(obs)))
    tot = 0.0; n = 0
    for it in range(4):
        t0 = time.perf_counter()
        for t in range(T):
            g._ck(g.lib.ppo_rollout_act(g.h, t, None, fp(act)))
            g._ck(g.lib.ppo_rollout_observe(g.h, t, fp(obs), fp(rew), fp(dn)))
        t1 = time.perf_counter()
        g._ck(g.lib.ppo_rollout_finish(g.h, C.c_float(0.99), C.c_float(0.95)))
        if it >= 1: tot += t1 - t0; n += T
    print("%.1f" % (1e6 * tot / n))
    sys.exit(0)
for E in [int(x) for x in sys.argv[1:]] or [1, 16, 64, 128, 256]:
    row = []
    for name, env in (("resident", {}), ("fused", {"PPO_HIP_NO_HOST_RESIDENT": "1"}), ("general", {"PPO_HIP_NO_HOST_FUSED": "1"})):
        out = subprocess.run([sys.executable, __file__, "--one", str(E)], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip().split("\n")[-1]
        row.append("%s %s us" % (name, out))
    print("E = %3d: " % E + " | ".join(row), flush=True)
