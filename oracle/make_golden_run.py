#!/usr/bin/env python3
"""Generates tests/golden/g45_run.npz: one complete, seeded pass of the hot path at the reference's shipped shape
(18 obs / 18 act, MLP [4,5], weights = the initialiser constants of the reference graph, tests/golden/g45_init.npz):
rollout over the seeded synthetic env -> GAE -> 2 epochs x 4 minibatches of train steps.

The arithmetic is the oracle's (oracle/ppo_oracle.c); before the file is written the losses and gradients of the first
minibatch are cross-checked against the independent float64 autograd restatement (oracle/torch_check.py), so the
fixture pins the oracle against drift and gives the HIP path a committed set of input/output vectors:

    inputs : noise [T,E,A], perms [epochs,B]              (+ the graph's initial weights, already a fixture)
    outputs: rollout obs/actions/values/neglogp/rewards/dones/returns, running statistics, loss rows [8,5],
             first-minibatch gradient + global norm, weights / Adam m / Adam v after the 8 train steps

    python oracle/make_golden_run.py        (needs nothing outside this repository)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as o          # noqa: E402
from oracle import torch_check as tc    # noqa: E402
from tests import helpers as H          # noqa: E402

E, T, NMB, EPOCHS = 8, 32, 4, 2
LR, CR, GAMMA, LAM = 0.000393141177482903, 0.16102319955825806, 0.99, 0.95


def main():
    orc = o.Oracle(18, 18, [4, 5])
    orc.set_tensors(H.g45_init())
    rng = np.random.RandomState(20190820)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, last_v = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    B = E * T
    perm = np.arange(B, dtype=np.int32); perms = []
    for _ in range(EPOCHS):
        rng.shuffle(perm); perms.append(perm.copy())
    perms = np.stack(perms)
    # first minibatch by hand: gradient + independent float64 cross-check
    flat = {k: np.ascontiguousarray(np.swapaxes(ro[k], 0, 1)).reshape((B,) + ro[k].shape[2:]) for k in
            ("obs", "actions", "values", "neglogp", "returns")}          # env-major rows r = e*T + t (runner.hpp:136-152)
    M = B // NMB
    shuf = {k: np.empty_like(v) for k, v in flat.items()}
    for k, v in flat.items():
        shuf[k][perms[0]] = v                                              # out.row(perm[i]) = in.row(i) (ppo2.hpp:291-296)
    mb = {k: v[:M] for k, v in shuf.items()}
    adv = o.adv_normalize(mb["returns"], mb["values"])
    losses0, grad0 = orc.loss_grad(mb["obs"], mb["actions"], adv, mb["returns"], mb["neglogp"], mb["values"], CR)
    ref_l, ref_g = tc.loss_and_grads(orc.named(), 2, mb["obs"], mb["actions"], adv, mb["returns"], mb["neglogp"], mb["values"], CR,
                                     o.G_ENT_COEF, o.G_VF_COEF)
    np.testing.assert_allclose(losses0, ref_l, rtol=2e-5, atol=1e-6)
    for name, g in orc.named(grad0).items():
        np.testing.assert_allclose(g, ref_g[name].reshape(g.shape), rtol=2e-4, atol=2e-6 * max(np.abs(x).max() for x in ref_g.values()))
    _, norm0 = orc.clip(grad0.copy())
    rows, mean = orc.update(ro, perms, NMB, LR, CR)
    np.testing.assert_allclose(rows[0], losses0, rtol=1e-6, atol=1e-7)
    out = dict(E=E, T=T, nmb=NMB, epochs=EPOCHS, lr=LR, cr=CR, gamma=GAMMA, lam=LAM, seed=1234, noise=noise, perms=perms,
               last_values=last_v, loss_rows=rows, loss_mean=mean, grad0=grad0, norm0=np.float32(norm0),
               theta=orc.theta.copy(), adam_m=orc.m.copy(), adam_v=orc.v.copy(), beta_pow=np.asarray(orc.pow, np.float32).copy(),
               obs_mean=nz.obs_rms.mean.copy(), obs_var=nz.obs_rms.var.copy(), obs_count=np.float64(nz.obs_rms.count),
               ret_mean=nz.ret_rms.mean.copy(), ret_var=nz.ret_rms.var.copy(), ret_count=np.float64(nz.ret_rms.count))
    out.update({"ro_" + k: v for k, v in ro.items()})
    path = os.path.join(ROOT, "tests", "golden", "g45_run.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss rows[0] =", rows[0])


if __name__ == "__main__":
    main()
