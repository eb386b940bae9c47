"""weight_grad_assemble_kernel's slab hand-off in its three forms -- the default (write-through slabs, relaxed arrival, sc1 loads), PPO_HIP_DW2_FENCES=1 (agent-scope release /
acquire) and PPO_HIP_DW2_OWN_LINES=1 (first-layer strips contiguous inside a slab) -- must give the same bits: N train steps of a [256,256] net per form, weights / Adam slots /
loss rows compared bitwise, and the time per step of each.  The two opt-in forms were written at the end of round 5 without a device to run them on: run this first.
usage: python tools/dw2_forms.py [steps] [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_cpp_amd
from oracle import oracle as o
from tests import helpers as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
LR, CR = 3.93141e-4, 0.161023


def run(env):
    for k in ("PPO_HIP_DW2_FENCES", "PPO_HIP_DW2_OWN_LINES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    orc = o.Oracle(18, 18, [256, 256]); orc.init_orthogonal(3)
    g = ppo_cpp_amd.PPOHip(18, 18, [256, 256]); g.set_flat(orc.theta.copy())
    mbs = [H.synth_minibatch(orc, ROWS, seed=70 + i) for i in range(4)]
    rows = []
    t0 = time.time()
    for it in range(N):
        mb = mbs[it % 4]
        rows.append(np.asarray(g.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])).copy())
    dt = (time.time() - t0) / N
    out = (g.get_flat(0), g.get_flat(1), g.get_flat(2), np.array(rows), g.debug_buffer("theta"), g.debug_buffer("thetaT"))
    counts = g.kernel_counts(); g.close()
    assert counts.get("weight_grad_assemble_kernel", 0) > 0, counts
    return out, dt


ref, t_ref = run({})
print("default: %.1f us per train step call (host-inclusive)" % (t_ref * 1e6), flush=True)
ok = True
for name, env in (("PPO_HIP_DW2_FENCES=1", {"PPO_HIP_DW2_FENCES": "1"}), ("PPO_HIP_DW2_OWN_LINES=1", {"PPO_HIP_DW2_OWN_LINES": "1"}),
                  ("both", {"PPO_HIP_DW2_FENCES": "1", "PPO_HIP_DW2_OWN_LINES": "1"})):
    got, dt = run(env)
    same = all(np.array_equal(a, b) for a, b in zip(ref, got))
    ok = ok and same
    print("%s: bitwise equal to the default over %d steps of %d rows: %s (%.1f us per call)" % (name, N, ROWS, same, dt * 1e6), flush=True)
assert ok
