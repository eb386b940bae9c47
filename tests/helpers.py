"""Shared test helpers: fixture loading and seeded synthetic minibatches."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz_named(fname):
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=True)
    return {k.replace("__", "/"): z[k] for k in z.files}


def g45_init():
    d = load_npz_named("g45_init.npz")
    return {k: v for k, v in d.items() if not k.startswith("const") and k != "adam_order"}


def g45_consts():
    d = load_npz_named("g45_init.npz")
    return dict(zip([str(x) for x in d["const_names"]], [float(x) for x in d["const_values"]])), \
        [str(x) for x in d["adam_order"]]


def ckpt71():
    return load_npz_named("ckpt71.npz")


def ckpt71_stats():
    return json.load(open(os.path.join(GOLDEN, "ckpt71_stats.json")))


def synth_minibatch(orc, n, seed, adv_normalized=True, cr=0.16102319955825806):
    """A seeded minibatch that exercises BOTH clip branches: old_neglogp/old_values are perturbed copies of the
    current model's outputs so that ratio and v - v_old straddle the clip range."""
    from oracle import oracle as o
    rng = np.random.RandomState(seed)
    obs = rng.uniform(-1, 1, (n, orc.O)).astype(np.float32)
    noise = rng.normal(size=(n, orc.A)).astype(np.float32)
    act, v, nlp = orc.step(obs, noise)
    old_nlp = (nlp + rng.normal(scale=0.15, size=n)).astype(np.float32)
    old_v = (v + rng.normal(scale=0.2, size=n)).astype(np.float32)
    ret = (v + rng.normal(scale=0.5, size=n)).astype(np.float32)
    # Keep every row CLEAR of the discontinuities of the loss (ratio == 1 +- cr, |v - v_old| == cr, (v-R)^2 == (vclip-R)^2):
    # within ~1e-5 of one of them, which side a row falls on is decided by the last bits of a 64-term fp32 sum, i.e. by
    # the summation order, and one flipped row moves the whole gradient by ~1/sqrt(n).  Rows inside a 1e-3 band are pushed
    # away (exact ties, ratio == 1 and v == v_old, stay: they are the common case and have their own deterministic rule).
    ratio = np.exp(old_nlp.astype(np.float64) - nlp)
    near = np.abs(np.abs(ratio - 1.0) - cr) < 1e-3
    old_nlp[near] += np.float32(0.01)
    dvo = v.astype(np.float64) - old_v
    near = np.abs(np.abs(dvo) - cr) < 1e-3
    old_v[near] -= np.float32(0.01) * np.sign(dvo[near]).astype(np.float32)
    dvo = v.astype(np.float64) - old_v
    vclip = old_v + np.clip(dvo, -cr, cr)
    s1, s2 = (v - ret.astype(np.float64)) ** 2, (vclip - ret) ** 2
    near = (np.abs(dvo) > cr) & (np.abs(s1 - s2) < 1e-3 * np.maximum(s1, 1e-6))
    ret[near] += np.float32(0.05)
    adv = o.adv_normalize(ret, old_v) if adv_normalized else (ret - old_v).astype(np.float32)
    return dict(obs=obs, actions=act, advs=adv, returns=ret, old_neglogp=old_nlp, old_values=old_v)


G_TENSORS = ["pi_fc0/w", "pi_fc0/b", "vf_fc0/w", "vf_fc0/b", "pi_fc1/w", "pi_fc1/b", "vf_fc1/w", "vf_fc1/b", "vf/w", "vf/b", "pi/w", "pi/b", "pi/logstd"]


def graph_run():
    """tests/golden/g45_graph_run.npz: outputs of the reference's graph file executed node by node
    (oracle/graph_interp.py + oracle/make_graph_golden.py); keys like 'train0/losses', 'train1/w:pi/w'."""
    return load_npz_named("g45_graph_run.npz")


def graph_state(z, prefix, kind):
    """{tensor name: array} of the weights ('w'), Adam m ('m') or Adam v ('v') stored under `prefix`."""
    return {t: z["%s/%s:%s" % (prefix, kind, t)] for t in G_TENSORS}


def golden_run(tag):
    """(z, hidden, stride, weights): a committed golden run (oracle/make_golden_run.py) and how to rebuild its initial weights:
    'g45' = the reference graph's own constants; 'g6464' / 'g256' = seeded weights (oracle/make_golden_run.seeded_weights)."""
    z = np.load(os.path.join(GOLDEN, tag + "_run.npz"))
    if tag == "g45":
        return z, (4, 5), 1, None
    return z, tuple(int(x) for x in z["hidden"]), int(z["stride"]), int(z["weight_seed"])


def golden_weights(orc, z_seed):
    """initial weights of a golden run into an oracle.Oracle (returns its flat theta)"""
    if z_seed is None:
        orc.set_tensors(g45_init())
    else:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
        from oracle import make_golden_run as mg
        mg.seeded_weights(orc, z_seed)
    return orc.theta
