#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's graph file (TEST INFRASTRUCTURE, build container only).

Runs oracle/graph_interp.py over G (= /root/reference/resources/ppo_cl/graphs/ppo_cpp_[4_5]_...meta.txt) with the
exact feeds / fetches / targets of the reference's Session::Run call sites and stores inputs and outputs in
tests/golden/g45_graph_run.npz:

  init                         session_creator.hpp:54   target "init"           -> the 15 model tensors, Adam slots, beta powers
  act (37 rows)                policies.hpp:33-77       feed input/Ob:0         -> output/_action, _deterministic_action, _value_flat, _neglogp
  3 train steps (64 rows)      ppo2.hpp:430-468         feed the 8 placeholders -> loss/pg_loss, vf_loss, ppo2/entropy, approxkl, clipfrac;
                                                        target ppo2/_train      -> weights, Adam m / v, beta powers after every step;
                               plus the 13 raw gradients (inputs of loss/global_norm/L2Loss*) and loss/global_norm/global_norm
  1 poisoned train step        an infinite advantage -> non-finite norm -> every trainable tensor NaN (G:24493-24543)

Feeds are seeded synthetic data (the reference ships no recorded inputs); old_neglogp / old_values are perturbed
copies of the graph's own act outputs so that ratio and v - v_old straddle the clip range (both Select branches and
the tie rules of every Maximum / Minimum run).  The minibatch advantages are normalised as ppo2.hpp:401-406 does on
the host (that step is not part of G).

The file holds numbers only.  Regenerate with:  python oracle/make_graph_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import graph_interp as gi  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
LR, CR = np.float32(0.000393141177482903), np.float32(0.16102319955825806)       # README.md:70-81
LOSSES = ["loss/pg_loss:0", "loss/vf_loss:0", "loss/ppo2/entropy:0", "loss/approxkl:0", "loss/clipfrac:0"]   # ppo2.hpp:539-543
ACT = ["output/_action:0", "output/_deterministic_action:0", "output/_value_flat:0", "output/_neglogp:0"]   # ppo2.hpp:523-526
NOISE = "output/random_normal/RandomStandardNormal"
TENSORS = ["pi_fc0/w", "pi_fc0/b", "vf_fc0/w", "vf_fc0/b", "pi_fc1/w", "pi_fc1/b", "vf_fc1/w", "vf_fc1/b", "vf/w", "vf/b", "pi/w", "pi/b", "pi/logstd"]


def adv_normalize(ret, val):
    """ppo2.hpp:401-406 (host side): (adv - mean) / (sqrt(mean((adv - mean)^2)) + 1e-8), float32 Eigen arithmetic."""
    adv = (ret - val).astype(np.float32)
    mean = np.float32(adv.sum(dtype=np.float32) / np.float32(adv.size))
    d = (adv - mean).astype(np.float32)
    std = np.float32(np.sqrt(np.float32((d * d).sum(dtype=np.float32) / np.float32(adv.size))))
    return (d / np.float32(np.float64(std) + 1e-8)).astype(np.float32)


def state(G):
    out = {}
    for t in TENSORS:
        out["w:" + t] = G.vars["model/" + t].copy()
        out["m:" + t] = G.vars["model/" + t + "/Adam"].copy()
        out["v:" + t] = G.vars["model/" + t + "/Adam_1"].copy()
    out["beta_pow"] = np.array([G.vars["beta1_power"], G.vars["beta2_power"]], np.float32)
    return out


def run(matmul_mode):
    G = gi.GraphInterp(gi.default_graph_path(), matmul_mode=matmul_mode)
    G.init()
    z = {}
    for k, v in state(G).items():
        z["init/" + k] = v
    # the gradient tensors in the order G stacks them for the global norm (= ApplyAdam order)
    l2 = ["loss/global_norm/L2Loss"] + ["loss/global_norm/L2Loss_%d" % i for i in range(1, 13)]
    grad_names = [G.inputs[n][0] for n in l2]
    adam = [n for n, op in G.ops.items() if op == "ApplyAdam"]
    assert [G.inputs[n][0] for n in adam] == ["model/" + t for t in TENSORS], "ApplyAdam order"
    rng = np.random.RandomState(20240)
    # ---- act ----
    obs = rng.uniform(-1, 1, (37, 18)).astype(np.float32)
    noise = rng.normal(size=(37, 18)).astype(np.float32)
    a, mu, v, nlp = G.run(ACT, {"input/Ob:0": obs}, noise={NOISE: noise})
    z.update({"act/obs": obs, "act/noise": noise, "act/action": a, "act/det_action": mu, "act/value": v, "act/neglogp": nlp})
    act_nodes = list(G.executed)
    # ---- train ----
    n = 64
    train_nodes = set()
    for step in range(3):
        tobs = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        tnoise = rng.normal(size=(n, 18)).astype(np.float32)
        ta, _, tv, tnlp = G.run(ACT, {"input/Ob:0": tobs}, noise={NOISE: tnoise})       # rollout with the CURRENT weights
        old_nlp = (tnlp + rng.normal(scale=0.15, size=n)).astype(np.float32)
        old_v = (tv + rng.normal(scale=0.2, size=n)).astype(np.float32)
        ret = (tv + rng.normal(scale=0.5, size=n)).astype(np.float32)
        if step == 1:
            old_nlp[:8] = tnlp[:8]; old_v[:8] = tv[:8]                                   # exact ties: ratio == 1, v == v_old
        adv = adv_normalize(ret, old_v)
        feeds = {"train_model/input/Ob:0": tobs, "loss/action_ph:0": ta, "loss/advs_ph:0": adv, "loss/rewards_ph:0": ret,
                 "loss/old_neglog_pac_ph:0": old_nlp, "loss/old_vpred_ph:0": old_v, "loss/learning_rate_ph:0": LR, "loss/clip_range_ph:0": CR}
        out = G.run(LOSSES + [g + ":0" if ":" not in g else g for g in grad_names] + ["loss/global_norm/global_norm:0"], feeds, targets=["ppo2/_train"])
        train_nodes |= set(G.executed)
        p = "train%d/" % step
        z.update({p + "obs": tobs, p + "actions": ta, p + "advs": adv, p + "returns": ret, p + "old_neglogp": old_nlp, p + "old_values": old_v})
        z[p + "losses"] = np.array([float(x) for x in out[:5]], np.float32)
        for t, g in zip(TENSORS, out[5:18]):
            z[p + "grad:" + t] = np.asarray(g, np.float32)
        z[p + "global_norm"] = np.float32(out[18])
        for k, val in state(G).items():
            z[p + k] = val
    # ---- non-finite gradient ----
    feeds = dict(feeds); bad = adv.copy(); bad[0] = np.inf; feeds["loss/advs_ph:0"] = bad
    G.run(LOSSES, feeds, targets=["ppo2/_train"])
    z["poison/all_nan"] = np.array([bool(np.isnan(G.vars["model/" + t]).all()) for t in TENSORS])
    census = {}
    for nm in set(act_nodes) | train_nodes:
        census[G.ops[nm]] = census.get(G.ops[nm], 0) + 1
    z["meta/op_census"] = np.array(sorted("%s=%d" % kv for kv in census.items()))
    z["meta/lr_cr"] = np.array([LR, CR], np.float32)
    return z


def main():
    a = run("f64round")
    b = run("f32chain")
    worst = 0.0
    for k in a:
        if a[k].dtype.kind != "f" or k.startswith("poison"):
            continue
        scale = max(float(np.abs(a[k]).max()), 1e-12)
        worst = max(worst, float(np.abs(a[k].astype(np.float64) - b[k]).max()) / scale)
    print("max deviation between the two MatMul accumulation orders, relative to each tensor's max: %.3g" % worst)
    assert worst < 5e-5, worst
    a["meta/matmul_order_spread"] = np.float32(worst)
    assert a["poison/all_nan"].all()
    path = os.path.join(GOLDEN, "g45_graph_run.npz")
    np.savez_compressed(path, **{k.replace("/", "__"): v for k, v in a.items()})
    print("wrote", path, os.path.getsize(path), "bytes;", len(a), "arrays")
    print("losses:", [a["train%d/losses" % i] for i in range(3)])
    print("ops executed:", " ".join(a["meta/op_census"]))


if __name__ == "__main__":
    main()
