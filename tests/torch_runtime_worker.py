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
    maps = open("/proc/self/maps").read()
    res["hip_runtimes"] = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
    json.dump(res, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()
