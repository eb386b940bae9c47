"""Pins the CPU oracle (oracle/ppo_oracle.c) against everything the reference tree offers for this path
(SURVEY 8c): the initial weights embedded in G, the trained checkpoint ...pkl.71 + its JSON running stats,
analytic known answers, and an independent torch-float64 autograd restatement.  CPU only."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import pytest

from oracle import oracle as o
from oracle import torch_check as tc
from tests import helpers as H

CR = 0.16102319955825806     # cliprange of the shipped run (ckpt71_stats.json)
LR = 0.000393141177482903


def make(hidden=(4, 5), O=18, A=18):
    return o.Oracle(O, A, list(hidden))


# ---------------------------------------------------------------------------------------------
# fixtures extracted from the reference's data files
# ---------------------------------------------------------------------------------------------
def test_graph_constants_match_oracle_defaults():
    consts, adam_order = H.g45_consts()
    assert consts["loss/mul_4/y"] == pytest.approx(o.G_ENT_COEF, rel=0, abs=0)
    assert consts["loss/mul_5/y"] == o.G_VF_COEF
    assert consts["loss/clip_by_global_norm/mul/x"] == o.G_MAX_GRAD_NORM
    assert consts["ppo2/_train/beta1"] == o.G_BETA1 and consts["beta1_power/initial_value"] == o.G_BETA1
    assert consts["ppo2/_train/beta2"] == o.G_BETA2 and consts["beta2_power/initial_value"] == o.G_BETA2
    assert consts["ppo2/_train/epsilon"] == o.G_EPS
    # the flat layout follows the ApplyAdam / global-norm order of G
    assert adam_order == [n for n, _, _ in make().tensors]


def test_graph_initial_weights_are_orthogonal():
    """a16: orthogonal init, gain sqrt2 (hidden) / 0.01 (pi, q) / 1.0 (vf); biases and logstd zero."""
    w = H.g45_init()
    for name, gain in (("pi_fc0/w", 2 ** 0.5), ("vf_fc0/w", 2 ** 0.5), ("pi_fc1/w", 2 ** 0.5), ("vf_fc1/w", 2 ** 0.5),
                       ("pi/w", 0.01), ("q/w", 0.01), ("vf/w", 1.0)):
        m = w[name].astype(np.float64)
        gram = m.T @ m if m.shape[0] >= m.shape[1] else m @ m.T
        np.testing.assert_allclose(gram, gain ** 2 * np.eye(gram.shape[0]), atol=2e-6 * max(1, gain ** 2))
    for name in ("pi_fc0/b", "vf_fc0/b", "pi_fc1/b", "vf_fc1/b", "pi/b", "vf/b", "pi/logstd", "q/b"):
        assert not w[name].any()


def test_initial_model_known_answers():
    """neglogp(a = mu) = 18*0.9189385 = 16.5409 and entropy = 18*1.4189385 = 25.5409 at logstd = 0."""
    orc = make()
    orc.set_tensors(H.g45_init())
    rng = np.random.RandomState(0)
    obs = rng.uniform(-1, 1, (7, 18)).astype(np.float32)
    a, v, nlp = orc.step(obs, np.zeros((7, 18), np.float32))
    mu, v2 = orc.forward(obs)
    np.testing.assert_array_equal(a, mu)
    np.testing.assert_array_equal(v, v2)
    np.testing.assert_allclose(nlp, 18 * 0.9189385175704956, rtol=1e-6)
    mb = H.synth_minibatch(orc, 64, 1)
    losses, _ = orc.loss_grad(cliprange=CR, **{k: mb[k] for k in ("obs", "actions", "advs", "returns")},
                              old_nlp=mb["old_neglogp"], old_v=mb["old_values"])
    assert losses[2] == pytest.approx(18 * 1.4189385175704956, rel=1e-6)
    # with sigma = 1, neglogp = 0.5*|noise|^2 + const
    noise = rng.normal(size=(7, 18)).astype(np.float32)
    _, _, nlp = orc.step(obs, noise)
    np.testing.assert_allclose(nlp, 0.5 * (noise.astype(np.float64) ** 2).sum(1) + 18 * 0.9189385175704956, rtol=2e-6)


def test_checkpoint_survey_sanity_values():
    """mu(0)[0..2] = -0.2594, -0.6368, -0.4799 ; V(0) = 2.5878 ; sigma in [0.31, 0.60]."""
    orc = make()
    ck = H.ckpt71()
    orc.set_tensors(ck)
    mu, v = orc.forward(np.zeros((1, 18), np.float32))
    np.testing.assert_allclose(mu[0, :3], [-0.2594, -0.6368, -0.4799], atol=5e-5)
    assert v[0] == pytest.approx(2.5878, abs=5e-5)
    sigma = np.exp(ck["pi/logstd"])
    assert 0.30 <= sigma.min() and sigma.max() <= 0.61     # survey quotes the range rounded to [0.31, 0.60]
    # independent numpy-float64 evaluation of the same checkpoint
    h = np.tanh(np.zeros((1, 18)) @ ck["pi_fc0/w"].astype(np.float64) + ck["pi_fc0/b"])
    h = np.tanh(h @ ck["pi_fc1/w"].astype(np.float64) + ck["pi_fc1/b"])
    np.testing.assert_allclose(mu[0], (h @ ck["pi/w"].astype(np.float64) + ck["pi/b"])[0], rtol=2e-6, atol=2e-7)


def test_checkpoint_json_running_stats_fixture():
    st = H.ckpt71_stats()
    assert st["n_steps"] == 65536 and st["nminibatches"] == 32 and st["noptepochs"] == 10
    assert st["obs_rms"]["count"] == pytest.approx(72001473.000001)
    assert len(st["obs_rms"]["mean"]) == 18 and len(st["obs_rms"]["var"]) == 18
    # normalising with frozen checkpoint stats (training=False): (x-mean)/sqrt(var+1e-8), clip 10
    nz = o.Normalizer(4, 18, training=False)
    nz.obs_rms.mean[:] = st["obs_rms"]["mean"]
    nz.obs_rms.var[:] = st["obs_rms"]["var"]
    nz.obs_rms.count = st["obs_rms"]["count"]
    x = np.random.RandomState(3).uniform(-3, 3, (4, 18)).astype(np.float32)
    ref = np.clip((x.astype(np.float64) - np.float32(st["obs_rms"]["mean"])) /
                  np.sqrt(np.float32(st["obs_rms"]["var"]).astype(np.float64) + 1e-8), -10, 10)
    np.testing.assert_allclose(nz.obs(x), ref, rtol=3e-6, atol=1e-6)
    assert np.abs(ref).max() == 10.0                      # the clip is exercised
    assert nz.obs_rms.count == pytest.approx(72001473.000001)   # frozen


# ---------------------------------------------------------------------------------------------
# independent autograd restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hidden,n,src", [((4, 5), 64, "ginit"), ((4, 5), 256, "ckpt"), ((64, 64), 128, "orth"),
                                          ((32,), 96, "orth"), ((16, 8, 8), 64, "orth")])
def test_loss_and_gradients_match_torch_autograd(hidden, n, src):
    orc = make(hidden)
    if src == "ginit":
        orc.set_tensors(H.g45_init())
    elif src == "ckpt":
        orc.set_tensors(H.ckpt71())
    else:
        orc.init_orthogonal(5)
        orc.tensor("pi/logstd")[:] = np.random.RandomState(9).uniform(-1.0, 0.2, (1, 18))
    mb = H.synth_minibatch(orc, n, seed=11)
    losses, grad = orc.loss_grad(mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"],
                                 mb["old_values"], CR)
    ref_losses, ref_grads = tc.loss_and_grads(orc.named(), len(hidden), mb["obs"], mb["actions"], mb["advs"],
                                              mb["returns"], mb["old_neglogp"], mb["old_values"], CR,
                                              o.G_ENT_COEF, o.G_VF_COEF)
    assert 0.05 < losses[4] < 0.95, "test data must exercise clipped and unclipped rows"
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5, atol=1e-6)
    gscale = max(np.abs(g).max() for g in ref_grads.values())
    for name, g in orc.named(grad).items():
        np.testing.assert_allclose(g, ref_grads[name].reshape(g.shape), rtol=2e-4, atol=2e-6 * gscale, err_msg=name)


def test_gradient_tie_rows_follow_first_argument():
    """Unclipped rows have m1 == m2 / s1 == s2 exactly: TF sends the gradient through the FIRST argument; the
    value must equal autograd's (split) convention.  old == current => every row is a tie."""
    orc = make((4, 5))
    orc.set_tensors(H.ckpt71())
    rng = np.random.RandomState(2)
    obs = rng.uniform(-1, 1, (32, 18)).astype(np.float32)
    a, v, nlp = orc.step(obs, rng.normal(size=(32, 18)).astype(np.float32))
    ret = (v + rng.normal(size=32)).astype(np.float32)
    adv = o.adv_normalize(ret, v)
    losses, grad = orc.loss_grad(obs, a, adv, ret, nlp, v, CR)
    assert losses[4] == 0.0 and losses[3] == 0.0
    _, ref = tc.loss_and_grads(orc.named(), 2, obs, a, adv, ret, nlp, v, CR, o.G_ENT_COEF, o.G_VF_COEF)
    for name, g in orc.named(grad).items():
        np.testing.assert_allclose(g, ref[name].reshape(g.shape), rtol=3e-4, atol=1e-6, err_msg=name)


def test_train_step_sequence_matches_float64_clip_and_adam():
    """5 consecutive train steps (loss -> backward -> global-norm clip -> TF ApplyAdam with beta powers starting
    at beta) against the float64 restatement."""
    hidden = (4, 5)
    orc = make(hidden)
    orc.set_tensors(H.g45_init())
    named = {k: v.astype(np.float64).copy() for k, v in orc.named().items()}
    m = {k: np.zeros_like(v) for k, v in named.items()}
    vv = {k: np.zeros_like(v) for k, v in named.items()}
    pw = (float(np.float32(o.G_BETA1)), float(np.float32(o.G_BETA2)))
    for it in range(5):
        mb = H.synth_minibatch(orc, 128, seed=100 + it)
        _, grads = tc.loss_and_grads(named, 2, mb["obs"], mb["actions"], mb["advs"], mb["returns"],
                                     mb["old_neglogp"], mb["old_values"], CR, o.G_ENT_COEF, o.G_VF_COEF)
        named, m, vv, pw, ref_norm = tc.clip_and_adam(named, grads, m, vv, pw, LR, o.G_MAX_GRAD_NORM,
                                                     o.G_BETA1, o.G_BETA2, o.G_EPS)
        _, norm, _ = orc.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], mb["returns"],
                                    mb["old_neglogp"], mb["old_values"])
        assert norm == pytest.approx(ref_norm, rel=2e-5)
        for k, w in orc.named().items():
            np.testing.assert_allclose(w, named[k].reshape(w.shape), rtol=1e-4, atol=2e-6, err_msg="%s @%d" % (k, it))
    assert orc.pow[0] == pytest.approx(0.9 ** 6, rel=1e-5) and orc.pow[1] == pytest.approx(0.999 ** 6, rel=1e-5)
    # first Adam step moves every weight with a non-zero gradient by ~lr (bias-corrected), a classic known answer
    fresh = make(hidden)
    fresh.set_tensors(H.g45_init())
    before = fresh.theta.copy()
    mb = H.synth_minibatch(fresh, 128, seed=100)
    _, _, g = fresh.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"],
                               mb["old_values"])
    moved = np.abs(fresh.theta - before)
    big = np.abs(g) > 0.05            # sqrt(v)=0.0316|g| >> eps=1e-5 there, so |step| = lr within 1 %
    assert big.sum() >= 3
    np.testing.assert_allclose(moved[big], LR, rtol=0.01)


def test_global_norm_clip_semantics():
    orc = make((4, 5))
    g = np.random.RandomState(0).normal(size=orc.P).astype(np.float32)
    clipped, norm = orc.clip(g)
    assert norm == pytest.approx(float(np.linalg.norm(g.astype(np.float64))), rel=1e-6)
    assert float(np.linalg.norm(clipped.astype(np.float64))) == pytest.approx(0.5, rel=1e-5)
    small = (g * 1e-3).astype(np.float32)
    c2, n2 = orc.clip(small)
    np.testing.assert_array_equal(c2, small * np.float32(0.5 * (1.0 / 0.5)))      # scale == 1 exactly
    bad = g.copy(); bad[3] = np.inf
    c3, n3 = orc.clip(bad)
    assert not np.isfinite(n3) and np.isnan(c3).all()                              # G:24493-24543 NaN poisoning


# ---------------------------------------------------------------------------------------------
# host-side numerics
# ---------------------------------------------------------------------------------------------
def test_gae_against_direct_float64_recursion():
    rng = np.random.RandomState(4)
    T, E = 37, 5
    rew, val = rng.normal(size=(T, E)), rng.normal(size=(T, E))
    dones = (rng.uniform(size=(T, E)) < 0.1).astype(np.float64)
    lv, ld = rng.normal(size=E), (rng.uniform(size=E) < 0.3).astype(np.float64)
    got = o.gae(rew, val, dones, lv, ld, 0.99, 0.95)
    adv = np.zeros((T, E)); last = np.zeros(E)
    g, lam = float(np.float32(0.99)), float(np.float32(0.95))
    for t in reversed(range(T)):
        nnt = 1 - (ld if t == T - 1 else dones[t + 1])
        nv = lv if t == T - 1 else val[t + 1]
        delta = rew[t] + g * nv * nnt - val[t]
        last = delta + g * lam * nnt * last
        adv[t] = last
    np.testing.assert_allclose(got, adv + val, rtol=2e-5, atol=2e-6)
    # known answer: zero rewards/values except the bootstrap => returns[t] = gamma^(T-t) * last_value
    z = np.zeros((4, 1))
    got = o.gae(z, z, z, np.ones(1), np.zeros(1), 0.5, 1.0)
    np.testing.assert_allclose(got[:, 0], [0.5 ** 4, 0.5 ** 3, 0.5 ** 2, 0.5], rtol=1e-6)
    # a done flag on the bootstrap cuts it
    got = o.gae(z, z, z, np.ones(1), np.ones(1), 0.5, 1.0)
    assert not got.any()


def test_running_statistics_chan_merge_equals_pooled_moments():
    rng = np.random.RandomState(6)
    rs = o.RunningStats(3)
    chunks = [rng.normal(loc=2.0, scale=3.0, size=(n, 3)).astype(np.float32) for n in (16, 1, 300, 64)]
    for c in chunks:
        rs.update(c)
    allx = np.concatenate(chunks).astype(np.float64)
    # the 1e-6 pseudo-count with mean 0 / var 1 is negligible
    np.testing.assert_allclose(rs.mean, allx.mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rs.var, allx.var(0), rtol=1e-5)
    assert rs.count == pytest.approx(381 + 1e-6)


def test_reward_normalisation_and_return_reset():
    nz = o.Normalizer(3, 2, gamma=0.9)
    r1 = nz.reward(np.array([1.0, 2.0, 3.0]), np.array([0.0, 1.0, 0.0]))
    # ret = [1,2,3]; ret_rms updated with it: var = pooled var of {1,2,3} (+ negligible prior)
    np.testing.assert_allclose(r1, np.array([1, 2, 3]) / np.sqrt(2.0 / 3.0 + 1e-8), rtol=1e-5)
    np.testing.assert_allclose(nz.ret, [1.0, 0.0, 3.0])                    # done env reset to 0 (env_normalize.hpp:88)
    nz.reward(np.array([1.0, 1.0, 1.0]), np.zeros(3))
    np.testing.assert_allclose(nz.ret, [1.9, 1.0, 3.7], rtol=1e-6)
    nz.training = False                                                     # frozen stats (env_normalize.hpp:76)
    big = nz.reward(np.array([1e6, -1e6, 0.0]), np.zeros(3))
    assert big[0] == 10.0 and big[1] == -10.0                              # clip_reward


def test_advantage_normalisation():
    rng = np.random.RandomState(8)
    ret, val = rng.normal(size=512).astype(np.float32), rng.normal(size=512).astype(np.float32)
    adv = o.adv_normalize(ret, val)
    d = ret.astype(np.float64) - val
    np.testing.assert_allclose(adv, (d - d.mean()) / (d.std() + 1e-8), rtol=2e-5, atol=2e-6)
    assert abs(float(adv.mean())) < 1e-6 and float(adv.std()) == pytest.approx(1.0, rel=1e-5)


def test_seeded_env_statistics_and_determinism():
    obs, rew, dn = o.seeded_env_step(1234, 0, 4096, 7, 18)
    obs2, rew2, dn2 = o.seeded_env_step(1234, 0, 4096, 7, 18)
    np.testing.assert_array_equal(obs, obs2)
    part, _, _ = o.seeded_env_step(1234, 100, 16, 7, 18)                   # env0 offset = sharding by env id
    np.testing.assert_array_equal(part, obs[100:116])
    assert -1.0 <= obs.min() and obs.max() < 1.0
    assert abs(float(obs.mean())) < 0.01 and float(obs.var()) == pytest.approx(1.0 / 3.0, rel=0.02)
    tot = np.mean([o.seeded_env_step(1234, 0, 4096, s, 18)[2].mean() for s in range(30)])
    assert tot == pytest.approx(1.0 / 300.0, rel=0.25)


def test_update_loop_equals_manual_minibatching():
    """orc_update (ppo2.hpp:264-335 restated) == hand-rolled: out.row(perm[i]) = in.row(i), env-major rows,
    contiguous slices, per-minibatch advantage normalisation."""
    E, T, nmb, epochs = 6, 8, 4, 2
    B, M = E * T, E * T // nmb
    a = make((4, 5)); a.set_tensors(H.g45_init())
    b = make((4, 5)); b.set_tensors(H.g45_init())
    rng = np.random.RandomState(12)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(a, nz, 77, T, rng.normal(size=(T, E, 18)).astype(np.float32), 0.99, 0.95)
    perm = np.arange(B, dtype=np.int32); perms = []
    for _ in range(epochs):
        rng.shuffle(perm); perms.append(perm.copy())                        # cumulative shuffle (ppo2.hpp:288)
    perms = np.stack(perms)
    rows, mean = a.update(ro, perms, nmb, LR, CR)
    flat = {k: np.swapaxes(ro[k], 0, 1).reshape((B,) + ro[k].shape[2:]) for k in
            ("obs", "actions", "values", "neglogp", "returns")}           # env-major flatten (runner.hpp:136-152)
    got = []
    for ep in range(epochs):
        shuf = {k: np.empty_like(v) for k, v in flat.items()}
        for k in flat:
            shuf[k][perms[ep]] = flat[k]
        for s in range(0, B, M):
            sl = {k: v[s:s + M] for k, v in shuf.items()}
            adv = o.adv_normalize(sl["returns"], sl["values"])
            l, _, _ = b.train_step(LR, CR, sl["obs"], sl["actions"], adv, sl["returns"], sl["neglogp"], sl["values"])
            got.append(l)
    np.testing.assert_array_equal(rows, np.stack(got))
    np.testing.assert_array_equal(a.theta, b.theta)
    np.testing.assert_allclose(mean, np.stack(got).astype(np.float64).mean(0), rtol=1e-6)


@pytest.mark.parametrize("tag", ["g45", "g6464", "g256"])
def test_committed_golden_run_is_reproduced(tag):
    """tests/golden/{g45,g6464,g256}_run.npz (oracle/make_golden_run.py): the reference's shipped [4,5] shape with the graph's
    initial weights, its real network shape [64,64] and BASELINE configs[2]'s [256,256] (seeded weights):
    rollout -> GAE -> 2 epochs x 4 minibatches.  The oracle must keep reproducing its committed vectors (libm differences
    between hosts allowed for: 1e-6)."""
    z, hidden, st, wseed = H.golden_run(tag)
    E, T, nmb = int(z["E"]), int(z["T"]), int(z["nmb"])
    orc = make(hidden); H.golden_weights(orc, wseed)
    nz = o.Normalizer(E, 18)
    ro, _, last_v = o.collect(orc, nz, int(z["seed"]), T, z["noise"], float(z["gamma"]), float(z["lam"]))
    for k in ("obs", "actions", "values", "neglogp", "rewards", "dones", "returns"):
        np.testing.assert_allclose(ro[k], z["ro_" + k], rtol=1e-6, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(nz.obs_rms.mean, z["obs_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nz.ret_rms.var, z["ret_var"], rtol=1e-6)
    assert nz.obs_rms.count == float(z["obs_count"])
    rows, mean = orc.update(ro, z["perms"], nmb, float(z["lr"]), float(z["cr"]))
    np.testing.assert_allclose(rows, z["loss_rows"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(orc.theta[::st], z["theta"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.m[::st], z["adam_m"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(orc.v[::st], z["adam_v"], rtol=1e-5, atol=1e-12)
    if tag == "g45":
        assert rows[0, 2] == pytest.approx(18 * 1.4189385, rel=1e-6) and rows[0, 3] == 0.0 and rows[0, 4] == 0.0
    else:
        assert np.sqrt(np.sum(orc.theta.astype(np.float64) ** 2)) == pytest.approx(float(z["theta_l2"]), rel=1e-6)


# ---- the oracle against the reference's graph file EXECUTED node by node -------------------------------------------
def _flat(orc, named):
    out = np.zeros(orc.P, np.float32)
    for n, off, shape in orc.tensors:
        out[off:off + int(np.prod(shape))] = np.asarray(named[n], np.float32).reshape(-1)
    return out


def test_oracle_matches_the_interpreted_graph():
    """tests/golden/g45_graph_run.npz holds what G itself computes (every node evaluated from its op, wiring and attrs
    by oracle/graph_interp.py; only TF's op-kernel semantics are restated there, no PPO formula).  The hand-written C
    restatement must reproduce it: act outputs, the five losses, all 13 raw gradients, the global norm, and weights /
    Adam slots / beta powers after each of three ApplyAdam rounds -- including exact-tie rows and both clip branches."""
    z = H.graph_run()
    spread = float(z["meta/matmul_order_spread"])          # how far two MatMul accumulation orders move G's own outputs
    assert spread < 5e-5
    orc = o.Oracle(18, 18, [4, 5])
    orc.set_tensors(H.graph_state(z, "init", "w"))
    np.testing.assert_array_equal(orc.theta, _flat(orc, H.g45_init()))          # the extractor and the interpreter agree on init
    np.testing.assert_array_equal(z["init/beta_pow"], orc.pow)
    a, v, nlp = orc.step(z["act/obs"], z["act/noise"])
    mu, _ = orc.forward(z["act/obs"])
    np.testing.assert_allclose(a, z["act/action"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mu, z["act/det_action"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, z["act/value"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nlp, z["act/neglogp"], rtol=1e-5, atol=1e-5)
    lr, cr = [float(x) for x in z["meta/lr_cr"]]
    for s in range(3):
        p = "train%d" % s
        _, grad = orc.loss_grad(z[p + "/obs"], z[p + "/actions"], z[p + "/advs"], z[p + "/returns"], z[p + "/old_neglogp"], z[p + "/old_values"], cr)
        losses, norm, _clipped = orc.train_step(lr, cr, z[p + "/obs"], z[p + "/actions"], z[p + "/advs"], z[p + "/returns"], z[p + "/old_neglogp"],
                                            z[p + "/old_values"])
        np.testing.assert_allclose(losses, z[p + "/losses"], rtol=1e-5, atol=1e-7, err_msg=p)
        assert losses[4] == z[p + "/losses"][4]                                   # clipfrac: a count, exact
        gref = _flat(orc, {t: z["%s/grad:%s" % (p, t)] for t in H.G_TENSORS})
        np.testing.assert_allclose(grad, gref, rtol=2e-5, atol=2e-6 * float(np.abs(gref).max()), err_msg=p + " gradient")
        assert norm == pytest.approx(float(z[p + "/global_norm"]), rel=1e-5)
        for kind, arr in (("w", orc.theta), ("m", orc.m), ("v", orc.v)):
            ref = _flat(orc, H.graph_state(z, p, kind))
            np.testing.assert_allclose(arr, ref, rtol=2e-5, atol=1e-7 if kind != "w" else 2e-6, err_msg="%s %s" % (p, kind))
        np.testing.assert_allclose(orc.pow, z[p + "/beta_pow"], rtol=1e-7)
    assert z["poison/all_nan"].all()
    bad = z["train2/advs"].copy(); bad[0] = np.inf
    orc.train_step(lr, cr, z["train2/obs"], z["train2/actions"], bad, z["train2/returns"], z["train2/old_neglogp"], z["train2/old_values"])
    assert np.isnan(orc.theta).all()                                              # G:24493-24543, as the graph itself does


def test_interpreted_graph_covers_the_hot_path_ops():
    """The golden run really executed the graph's train op: 13 ApplyAdam, 13 L2Loss, 4 TanhGrad, 22 MatMul (8 forward of
    the act model, 6 of the train model... as G wires them), the IsFinite guard, and no summary / saver node."""
    z = H.graph_run()
    census = dict(kv.split("=") for kv in z["meta/op_census"])
    assert census["ApplyAdam"] == "13" and census["L2Loss"] == "13" and census["TanhGrad"] == "4" and census["IsFinite"] == "1"
    assert int(census["MatMul"]) >= 20 and "ScalarSummary" not in census and "SaveV2" not in census


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only mounted in the build container")
def test_graph_golden_regenerates_from_the_reference_graph():
    """Build-container check that the committed vectors are what the interpreter produces from G today."""
    import subprocess, sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_graph_golden as mg
    fresh = mg.run("f64round")
    z = H.graph_run()
    for k, v in fresh.items():
        if k.startswith("meta/op_census"):
            continue
        np.testing.assert_array_equal(np.asarray(v), z[k], err_msg=k)


@pytest.mark.parametrize("hidden,n,O,A", [((64, 64), 256, 18, 18), ((256, 256), 300, 18, 18), ((32, 16), 64, 36, 7)])
def test_vectorised_numpy_port_equals_the_c_oracle(hidden, n, O, A):
    """oracle/numpy_port.py (bench.py's CPU baseline legs) computes what the C restatement computes: policy step, losses, gradient,
    clipped norm, three Adam rounds, GAE -- to fp32 rounding (BLAS summation order against double accumulators)."""
    from oracle import numpy_port as npp
    orc = o.Oracle(O, A, list(hidden)); orc.init_orthogonal(3)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(4).uniform(-1.0, 0.2, (1, A))
    twin = o.Oracle(O, A, list(hidden)); twin.theta[:] = orc.theta
    P = npp.NumpyPPO(twin)
    for it in range(3):
        mb = H.synth_minibatch(orc, n, seed=50 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, ref_grad = orc.loss_grad(*args, 0.161)
        _, ref_norm = orc.clip(ref_grad)
        orc.train_step(3.9e-4, 0.161, *args)
        losses, norm = P.train_step(3.9e-4, 0.161, *args)
        np.testing.assert_allclose(losses[:4], ref_losses[:4], rtol=1e-4, atol=1e-6)
        assert abs(float(losses[4]) - float(ref_losses[4])) <= 1.01 / n
        np.testing.assert_allclose(P.grad, ref_grad, rtol=2e-4, atol=2e-6 * float(np.abs(ref_grad).max()))
        assert norm == pytest.approx(ref_norm, rel=1e-4)
        np.testing.assert_allclose(twin.theta, orc.theta, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(twin.v, orc.v, rtol=4e-4, atol=1e-10)
    np.testing.assert_allclose(twin.pow, orc.pow, rtol=1e-6)
    rng = np.random.RandomState(1)
    obs = rng.uniform(-1, 1, (n, O)).astype(np.float32); noise = rng.normal(size=(n, A)).astype(np.float32)
    for x, y in zip(P.step(obs, noise), twin.step(obs, noise)):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-5)
    rw = rng.normal(size=(16, 8)).astype(np.float32); va = (0.3 * rw).astype(np.float32); dn = (rng.rand(16, 8) < 0.1).astype(np.float32)
    np.testing.assert_allclose(npp.gae(rw, va, dn, va[0], dn[0], 0.99, 0.95), o.gae(rw, va, dn, va[0], dn[0], 0.99, 0.95), rtol=1e-6, atol=1e-6)
    st = o.RunningStats(O)
    st.update(obs)
    m, v, c = npp.running_update(np.zeros(O, np.float32), np.ones(O, np.float32), 1e-6, obs)
    np.testing.assert_allclose(m, st.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(v, st.var, rtol=1e-5); assert c == st.count
