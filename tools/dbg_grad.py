#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of one train step at [256,256], n=2048 against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as o
from tests import helpers as H
import ppo_cpp_amd
hidden = [int(x) for x in os.environ.get("HIDDEN", "256,256").split(",")]; n = int(os.environ.get("N", "2048"))
orc = o.Oracle(18, 18, hidden); orc.init_orthogonal(3)
g = ppo_cpp_amd.PPOHip(18, 18, hidden); g.set_flat(orc.theta)
for it in range(2):
    mb = H.synth_minibatch(orc, n, seed=50 + it)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    ref_losses, ref_grad = orc.loss_grad(*args, 0.161)
    _, ref_norm = orc.clip(ref_grad)
    losses = g.train_step(3.9e-4, 0.161, *args)
    orc.train_step(3.9e-4, 0.161, *args)
    grad, norm = g.last_grad()
    print("it", it, "losses", losses, ref_losses, "norm", norm, ref_norm)
    for name, off, shape in orc.tensors:
        cnt = int(np.prod(shape))
        a, b = grad[off:off + cnt], ref_grad[off:off + cnt]
        err = np.abs(a - b).max(); print("  %-12s %-12s maxerr %.3e  ref max %.3e  nbad %d" % (name, shape, err, np.abs(b).max(), (np.abs(a - b) > 1e-6 + 2e-4 * np.abs(b)).sum()))
        if err > 1e-5 and len(shape) == 2:
            bad = np.argwhere(np.abs(a - b).reshape(shape) > 1e-6 + 2e-4 * np.abs(b).reshape(shape))
            print("     bad rows", np.unique(bad[:, 0])[:40], "cols", np.unique(bad[:, 1])[:40])
