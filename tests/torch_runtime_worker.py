"""One process of tests/test_race_guards.py::test_graph_replays_under_the_hip_runtime_bundled_with_torch; not collected by pytest.
usage: torch_runtime_worker.py <out.json>

`import torch` FIRST: the PyTorch wheel bundles its own libamdhip64.so (HIP 7.0.51831 in this image, against the system's 7.2.26015) under the same soname, so whichever is
loaded first serves the whole process -- libppo_hip.so included.  That is the situation of every pytest run that collects a module importing torch, and of bench.py's ranks;
it is also the TRIGGER of round 5's open finding: on that runtime a memset node at the head of the update's captured graph replays out of order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
from tests import test_other_shapes as t  # noqa: E402


def main():
    res = {"torch": torch.__version__}
    for name, members in (("beside the narrow handle", (0, 1)), ("alone", (1,))):
        seen = {}

        def counters(hs, i, it, dbg):
            if i == 1:
                seen[it] = int(np.count_nonzero(hs[1].debug_buffer("dw2_counters")))
        out = t._il_run(members, iterations=6, perms=True, after_update=counters)[0][1]
        errs = []
        for it in range(1, 6):
            rows, theta = t._il_oracle_leg(out, it)
            errs.append(float(np.max(np.abs(out[len(t._IL_NAMES) * it + 7] - theta))))
        res[name] = {"nonzero_counters": [seen[k] for k in sorted(seen)], "weight_err": errs}
    # the bf16 path's bucketed exchange forks and joins a second stream INSIDE the captured graph (events): under a one-rank communicator over the real collective library,
    # forced on (ppo_dist_bucketed 2), it must give what the single all-reduce gives up to the rounding of a different partial-sum split
    import ppo_cpp_amd
    outs = []
    for mode in (0, 2):
        g = ppo_cpp_amd.PPOHip(64, 20, [256, 256], compute_dtype=1)
        g.init_orthogonal(5)
        g.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
        g.dist_bucketed(mode)
        g.norm_init(64, 0.99); g.rollout_alloc(64, 16)
        g.collect_synthetic(7, 0.99, 0.95, None, env0=0, step0=0, first=True)
        for u in range(3):
            rows, _ = g.update(3.93141e-4, 0.161023, 2, 1, None, seed=u)
        outs.append((rows.copy(), g.get_flat(0)))
        res["bucketed %d graph collectives" % mode] = bool(g.dist_graph_collectives())
        g.close()
    res["bucketed_vs_single_rows_maxdiff"] = float(np.max(np.abs(outs[0][0] - outs[1][0])))
    res["bucketed_vs_single_theta_maxdiff"] = float(np.max(np.abs(outs[0][1] - outs[1][1])))
    maps = open("/proc/self/maps").read()
    res["hip_runtimes"] = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
    json.dump(res, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()
