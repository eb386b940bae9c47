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

g6464_run.npz / g256_run.npz: the same pass at the reference's real network shape [64,64] (ppo2.cpp:114) and at BASELINE
configs[2]'s [256,256], from seeded weights the tests rebuild (the reference ships constants only for [4,5]); the large
vectors are stored as strided samples plus whole-vector norms.

    python oracle/make_golden_run.py [g45] [g6464] [g256]       (needs nothing outside this repository)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as o          # noqa: E402
from oracle import torch_check as tc    # noqa: E402
from tests import helpers as H          # noqa: E402

LR, CR, GAMMA, LAM = 0.000393141177482903, 0.16102319955825806, 0.99, 0.95


def seeded_weights(orc, seed):
    """Deterministic non-trivial weights for shapes the reference ships no constants for: legacy RandomState streams are
    stable across NumPy versions, so the tests rebuild the same vector from the seed (nothing to store)."""
    rng = np.random.RandomState(seed)
    for n, off, shape in orc.tensors:
        cnt = int(np.prod(shape))
        if n.endswith("/w"):
            w = rng.normal(size=shape) / np.sqrt(shape[0])
            if n == "pi/w":
                w *= 0.05
        elif n == "pi/logstd":
            w = rng.uniform(-1.0, 0.2, size=shape)
        else:
            w = rng.normal(scale=0.05, size=shape)
        orc.theta[off:off + cnt] = w.astype(np.float32).reshape(-1)


def make(hidden, fname, E, T, nmb, epochs, stride, weight_seed=None):
    orc = o.Oracle(18, 18, list(hidden))
    if weight_seed is None:
        orc.set_tensors(H.g45_init())
    else:
        seeded_weights(orc, weight_seed)
    rng = np.random.RandomState(20190820)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, last_v = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    B = E * T
    perm = np.arange(B, dtype=np.int32); perms = []
    for _ in range(epochs):
        rng.shuffle(perm); perms.append(perm.copy())
    perms = np.stack(perms)
    # first minibatch by hand: gradient + independent float64 cross-check
    flat = {k: np.ascontiguousarray(np.swapaxes(ro[k], 0, 1)).reshape((B,) + ro[k].shape[2:]) for k in
            ("obs", "actions", "values", "neglogp", "returns")}          # env-major rows r = e*T + t (runner.hpp:136-152)
    M = B // nmb
    shuf = {k: np.empty_like(v) for k, v in flat.items()}
    for k, v in flat.items():
        shuf[k][perms[0]] = v                                              # out.row(perm[i]) = in.row(i) (ppo2.hpp:291-296)
    mb = {k: v[:M] for k, v in shuf.items()}
    adv = o.adv_normalize(mb["returns"], mb["values"])
    losses0, grad0 = orc.loss_grad(mb["obs"], mb["actions"], adv, mb["returns"], mb["neglogp"], mb["values"], CR)
    ref_l, ref_g = tc.loss_and_grads(orc.named(), len(hidden), mb["obs"], mb["actions"], adv, mb["returns"], mb["neglogp"], mb["values"], CR,
                                     o.G_ENT_COEF, o.G_VF_COEF)
    np.testing.assert_allclose(losses0, ref_l, rtol=2e-5, atol=1e-6)
    for name, g in orc.named(grad0).items():
        np.testing.assert_allclose(g, ref_g[name].reshape(g.shape), rtol=2e-4, atol=2e-6 * max(np.abs(x).max() for x in ref_g.values()))
    _, norm0 = orc.clip(grad0.copy())
    rows, mean = orc.update(ro, perms, nmb, LR, CR)
    np.testing.assert_allclose(rows[0], losses0, rtol=1e-6, atol=1e-7)
    st = stride
    out = dict(E=E, T=T, nmb=nmb, epochs=epochs, lr=LR, cr=CR, gamma=GAMMA, lam=LAM, seed=1234, noise=noise, perms=perms,
               last_values=last_v, loss_rows=rows, loss_mean=mean, grad0=grad0[::st], norm0=np.float32(norm0),
               theta=orc.theta[::st].copy(), adam_m=orc.m[::st].copy(), adam_v=orc.v[::st].copy(), beta_pow=np.asarray(orc.pow, np.float32).copy(),
               obs_mean=nz.obs_rms.mean.copy(), obs_var=nz.obs_rms.var.copy(), obs_count=np.float64(nz.obs_rms.count),
               ret_mean=nz.ret_rms.mean.copy(), ret_var=nz.ret_rms.var.copy(), ret_count=np.float64(nz.ret_rms.count))
    if weight_seed is not None:
        # strided samples + whole-vector checksums (the full [256,256] state would be megabytes)
        out.update(stride=st, weight_seed=weight_seed, hidden=np.array(hidden, np.int32),
                   theta_l2=np.float64(np.sqrt(np.sum(orc.theta.astype(np.float64) ** 2))), theta_sum=np.float64(orc.theta.astype(np.float64).sum()),
                   grad0_l2=np.float64(np.sqrt(np.sum(grad0.astype(np.float64) ** 2))))
        keep = ("obs", "actions", "values", "neglogp", "rewards", "dones", "returns")
        out.update({"ro_" + k: ro[k] for k in keep})
    else:
        out.update({"ro_" + k: v for k, v in ro.items()})
    path = os.path.join(ROOT, "tests", "golden", fname)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss rows[0] =", rows[0])


def main():
    which = sys.argv[1:] or ["g45", "g6464", "g256"]
    if "g45" in which:
        make((4, 5), "g45_run.npz", 8, 32, 4, 2, 1)
    if "g6464" in which:
        make((64, 64), "g6464_run.npz", 8, 32, 4, 2, 1, weight_seed=64)           # the reference's real shape (ppo2.cpp:114)
    if "g256" in which:
        make((256, 256), "g256_run.npz", 16, 16, 4, 2, 16, weight_seed=256)       # BASELINE configs[2]'s network


if __name__ == "__main__":
    main()
