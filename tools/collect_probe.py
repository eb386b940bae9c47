"""Per-kernel device times (HIP events) of one rollout collect; usage: python tools/collect_probe.py [E T O A H...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppo_cpp_amd
a = [int(x) for x in sys.argv[1:]]
E, T, O, A = (a + [4096, 16, 18, 18])[:4] if len(a) < 4 else a[:4]
hidden = a[4:] or [256, 256]
g = ppo_cpp_amd.PPOHip(O, A, hidden); g.init_orthogonal(0); g.norm_init(E, 0.99); g.rollout_alloc(E, T)
g.collect_synthetic(1, 0.99, 0.95, None, env0=0, step0=0, first=True)
for rep in range(2):
    g.sync(); t0 = time.perf_counter()
    g.collect_synthetic(1, 0.99, 0.95, None, env0=0, step0=T * (rep + 1), first=False)
    print("collect wall %.3f ms" % (1e3 * (time.perf_counter() - t0)))
g.prof_enable(True)
g.collect_synthetic(1, 0.99, 0.95, None, env0=0, step0=T * 5, first=False)
for k, (ms, n) in g.prof_read().items():
    if n: print("  %-14s %8.2f us x %d" % (k, 1e3 * ms / n, n))
