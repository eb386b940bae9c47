"""One rank of the two-process data-parallel test (tests/test_dp_two_ranks.py); not collected by pytest.
usage: dp_worker.py <rank> <world> <in.npz> <out.npz>      (PPO_RCCL_LIBRARY selects the collective library)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, fin, fout = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import ppo_cpp_amd
    d = np.load(fin)
    hidden = [int(x) for x in d["hidden"]]
    E, T, nmb, epochs = (int(d[k]) for k in ("E", "T", "nmb", "epochs"))
    El = E // world
    sl = slice(rank * El, (rank + 1) * El)
    O = int(d["O"]) if "O" in d.files else 18                      # (observation width: 64 / 256 exercise the column-group statistics under data parallelism)
    g = ppo_cpp_amd.PPOHip(O, 18, hidden, device=0, compute_dtype=1 if os.environ.get("PPO_TEST_BF16") == "1" else 0)
    g.set_flat(d["theta"])
    g.dist_init(world, rank, d["uid"].tobytes())
    if os.environ.get("PPO_TEST_PEER") == "1":
        # one-shot peer all-reduce: the 64-byte IPC handles travel through files next to the inputs (the launcher's control
        # plane in bench.py is gloo); ppo_dist_peer_attach itself ends with a collective, so nobody runs ahead
        import time
        base = os.path.dirname(os.path.abspath(fout))
        with open(os.path.join(base, "peer%d.tmp" % rank), "wb") as f:
            f.write(g.dist_peer_export())
        os.replace(os.path.join(base, "peer%d.tmp" % rank), os.path.join(base, "peer%d.bin" % rank))
        handles = []
        for r in range(world):
            fn = os.path.join(base, "peer%d.bin" % r)
            t0 = time.time()
            while not os.path.exists(fn):
                if time.time() - t0 > 120:
                    raise RuntimeError("no IPC handle from rank %d" % r)
                time.sleep(0.01)
            handles.append(open(fn, "rb").read())
        if os.environ.get("PPO_HIP_PEER_REDUCE") == "0":                    # the switch: regions exported and attached, the exchange stays on the collective library
            assert not g.dist_peer_attach(handles) and not g.dist_peer_active()
        else:
            if not g.dist_peer_attach(handles):
                raise RuntimeError("peer all-reduce probe failed (rank %d)" % rank)
            assert g.dist_graph_collectives()
    if os.environ.get("PPO_TEST_BUCKETED") == "0":
        g.dist_bucketed(False)                                          # (bf16 path: one all-reduce of the whole gradient instead of the layer buckets)
    g.norm_init(El, float(d["gamma"]))
    g.rollout_alloc(El, T)
    if os.environ.get("PPO_TEST_DRILL") == "1" and rank == world - 1:
        # failure drill: this rank never takes part in a collective again (it stays alive for a while, like a hung peer, then leaves)
        import time
        time.sleep(float(os.environ.get("PPO_TEST_DRILL_SLEEP", "4")))
        g.close()
        return
    iters = int(os.environ.get("PPO_TEST_ITERS", "0"))
    if iters:
        # determinism soak (tests/test_race_guards.py): many collect + update iterations with the library's own generators; everything that
        # crosses ranks (statistics table per env step, advantage moments per epoch, gradient per minibatch) is exercised `iters` times over
        means = []
        for i in range(iters):
            g.collect_synthetic(int(d["seed"]), float(d["gamma"]), float(d["lam"]), None, env0=rank * El, step0=i * T, first=(i == 0))
            means.append(g.update(float(d["lr"]), float(d["cr"]), epochs, nmb, None, seed=500 + i, want_rows=False)[1].copy())
        out = {"means": np.array(means), "theta": g.get_flat(0), "adam_m": g.get_flat(1), "adam_v": g.get_flat(2), "ro_returns": g.rollout_get("returns")}
        for which, nm in ((0, "obs"), (1, "ret")):
            m, v, c = g.norm_stats(which)
            out[nm + "_mean"], out[nm + "_var"], out[nm + "_count"] = m, v, np.float64(c)
        out["peer"] = np.int32(g.dist_peer_active())
        g.close()
        np.savez(fout, **out)
        return
    g.collect_synthetic(int(d["seed"]), float(d["gamma"]), float(d["lam"]), d["noise"][:, sl], env0=rank * El, step0=0, first=True)
    out = {"ro_" + f: g.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones")}
    for which, nm in ((0, "obs"), (1, "ret")):
        m, v, c = g.norm_stats(which)
        out[nm + "_mean"], out[nm + "_var"], out[nm + "_count"] = m, v, np.float64(c)
    for f in ("obs", "actions", "values", "neglogp", "returns"):          # identical inputs: isolate the update arithmetic
        g.rollout_set(f, d["ref_" + f][:, sl])
    if os.environ.get("PPO_TEST_GLOBAL") == "1":
        g.dist_global_shuffle(True)                                       # ONE permutation over the rows of all ranks, the same on every rank
        rows, mean = g.update(float(d["lr"]), float(d["cr"]), epochs, nmb, d["gperms"])
    else:
        rows, mean = g.update(float(d["lr"]), float(d["cr"]), epochs, nmb, d["perms"][rank])
    out["rows"], out["mean"], out["theta"] = rows, mean, g.get_flat(0)
    out["adam_m"], out["adam_v"] = g.get_flat(1), g.get_flat(2)
    out["comm_nranks"] = np.int32(g.dist_info()["comm_nranks"])
    g.close()
    np.savez(fout, **out)


if __name__ == "__main__":
    main()
