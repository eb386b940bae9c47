"""PPO_HIP_DEBUG_SNAPSHOT=<n> (read at ppo_create): every ppo_update copies all of the handle's device buffers into an arena right behind its train step n, readable through
ppo_debug_buffer as "snap:<name>".  Checks on [256,256] handles (launch per train step, update replayed from its graph; 256- and 1024-row minibatches):
the snapshot behind the LAST step holds the final weights and all loss rows; the one behind step 0 holds loss row 0, not yet row 1, and weights that are one Adam step from the
start; and the update's results are bitwise the same with and without the snapshot.  Written at the end of round 5 without a device to run it on: run it before trusting the
interleaved-handles report's snapshot section.  usage: python tools/snapshot_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppo_cpp_amd

LR, CR, GAMMA, LAM = 3.93141e-4, 0.161023, 0.99, 0.95


def run(hidden, E, T, nmb, snap):
    os.environ.pop("PPO_HIP_DEBUG_SNAPSHOT", None)
    if snap is not None:
        os.environ["PPO_HIP_DEBUG_SNAPSHOT"] = str(snap)
    g = ppo_cpp_amd.PPOHip(18, 18, list(hidden)); g.init_orthogonal(2); g.norm_init(E); g.rollout_alloc(E, T)
    out = []
    for it in range(2):                                                                  # the second update replays the graph
        g.collect_synthetic(41, GAMMA, LAM, None, step0=it * T, first=(it == 0))
        before = g.debug_buffer("theta")
        rows, mean = g.update(LR, CR, 2, nmb, None, seed=it)
        out.append((rows.copy(), g.debug_buffer("theta"), before,
                    g.debug_buffer("snap:theta") if snap is not None else None, g.debug_buffer("snap:loss_rows") if snap is not None else None))
    g.close()
    return out


for hidden, E, T, nmb in (((256, 256), 64, 16, 4), ((256, 256), 128, 16, 2)):     # (on the [64,64] shapes a step's clip + Adam rides in the NEXT launch: a snapshot there precedes it)
    steps = 2 * nmb
    plain = run(hidden, E, T, nmb, None)
    first = run(hidden, E, T, nmb, 0)
    last = run(hidden, E, T, nmb, steps - 1)
    for it in range(2):
        rows, theta, before = plain[it][:3]
        for other, name in ((first, "snapshot behind step 0"), (last, "snapshot behind the last step")):
            assert np.array_equal(other[it][0], rows) and np.array_equal(other[it][1], theta), "%s changes the update's results (update %d)" % (name, it)
        s_theta, s_rows = first[it][3], first[it][4].view(np.float32)
        assert s_theta.size == theta.size and not np.array_equal(s_theta, before) and not np.array_equal(s_theta, theta), "step 0's weights: neither the start nor the end"
        assert np.array_equal(s_rows[:5], rows[0]), "step 0's loss row"
        if it == 0:
            assert not s_rows[5:10].any(), "row 1 is not written yet behind step 0 of the first update"
        l_theta, l_rows = last[it][3], last[it][4].view(np.float32)
        assert np.array_equal(l_theta, theta), "the last step's weights are the final ones"
        assert np.array_equal(l_rows[:rows.size], rows.ravel()), "the last step's loss rows are all of them"
    print("hidden %s, %d x %d rows, %d minibatches: snapshots behind step 0 and step %d consistent, results unchanged: True" % (list(hidden), E, T, nmb, steps - 1), flush=True)
