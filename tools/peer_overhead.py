"""Cost of the data-parallel exchange machinery on ONE GPU (one-rank communicator: no link time, only the launches and the
local protocol): update phase per train step without a communicator, with ncclAllReduce (graph-captured when the probe passes)
and with the one-shot peer all-reduce; bf16 configurations (cfg5) also with the BUCKETED gradient exchange (ppo_dist_bucketed: a backward link, a weight-gradient launch, an
assembly and an all-reduce per layer instead of the chained backward + one launch of each) forced under the one-rank communicator -- what the per-layer launches cost one rank;
what the overlap buys needs more than one device.    usage: python tools/peer_overhead.py [cfg3|cfg4|cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ppo_cpp_amd

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = bench.CONFIGS[name]
E, T, nmb, ep = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"], cfg["noptepochs"]
for mode in ("none", "rccl", "peer") + (("rccl+buckets",) if cfg.get("dtype") == "bf16" else ()):
    g = ppo_cpp_amd.PPOHip(cfg["obs"], cfg["act"], cfg["hidden"], compute_dtype=1 if cfg.get("dtype") == "bf16" else 0)
    g.init_orthogonal(0)
    if mode != "none":
        g.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
    if mode == "peer":
        assert g.dist_peer_attach([g.dist_peer_export()])
    if mode == "rccl+buckets":
        g.dist_bucketed(2)
    g.norm_init(E, bench.GAMMA); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, bench.GAMMA, bench.LAM, None, env0=0, step0=0, first=True)
    for i in range(2):
        g.update(bench.LR, bench.CR, ep, nmb, None, seed=i, want_rows=False)
    g.sync(); t0 = time.perf_counter()
    for i in range(3):
        g.update(bench.LR, bench.CR, ep, nmb, None, seed=5 + i, want_rows=False)
    g.sync(); dt = (time.perf_counter() - t0) / 3
    t1 = time.perf_counter()
    for i in range(3):
        g.collect_synthetic(1234, bench.GAMMA, bench.LAM, None, env0=0, step0=(i + 1) * T, first=False)
    g.sync(); dc = (time.perf_counter() - t1) / 3
    print("%s %-12s update %.3f ms = %.1f us per train step (graph collectives: %s) ; collect %.3f ms = %.1f us per env step" %
          (name, mode, 1e3 * dt, 1e6 * dt / (ep * nmb), g.dist_graph_collectives() if mode != "none" else "-", 1e3 * dc, 1e6 * dc / T), flush=True)
    g.close()
