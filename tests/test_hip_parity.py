"""GPU parity tests: the HIP path (through the C-ABI, ctypes) against the CPU oracle on the same seeded inputs.
fp32 tolerances (north star: losses within 1e-4 rel; BASELINE.md section 4: 1e-5 abs / 1e-4 rel)."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu

CR = 0.16102319955825806
LR = 0.000393141177482903
GAMMA, LAM = 0.99, 0.95


def hip(hidden, O=18, A=18):
    import ppo_cpp_amd
    return ppo_cpp_amd.PPOHip(O, A, list(hidden))


def pair(hidden, src="orth", O=18, A=18, seed=3):
    orc = o.Oracle(O, A, list(hidden))
    if src == "ginit":
        orc.set_tensors(H.g45_init())
    elif src == "ckpt":
        orc.set_tensors(H.ckpt71())
    else:
        orc.init_orthogonal(seed)
        orc.tensor("pi/logstd")[:] = np.random.RandomState(seed + 1).uniform(-1.0, 0.2, (1, A))
    g = hip(hidden, O, A)
    g.set_flat(orc.theta)
    return orc, g


def close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def test_parameter_roundtrip_and_layout_order():
    orc, g = pair((4, 5), "ginit")
    assert g.P == orc.P == 334
    assert [n for n, _ in g.tensors] == [n for n, _, _ in orc.tensors]
    np.testing.assert_array_equal(g.get_flat(), orc.theta)
    np.testing.assert_array_equal(g.get_tensor("pi_fc0/w"), H.g45_init()["pi_fc0/w"])
    np.testing.assert_array_equal(g.beta_powers(), np.float32([0.9, 0.999]))
    g2 = hip((256, 256))
    assert g2.P == 146213
    g2.init_orthogonal(0)
    w = g2.get_tensor("pi_fc1/w").astype(np.float64)
    close(w.T @ w, 2 * np.eye(256), atol=1e-5)
    w = g2.get_tensor("pi/w").astype(np.float64)
    close(w.T @ w, 1e-4 * np.eye(18), atol=1e-9)
    assert not g2.get_tensor("pi/logstd").any()


@pytest.mark.parametrize("hidden,n,src", [((4, 5), 1, "ginit"), ((4, 5), 37, "ckpt"), ((64, 64), 100, "orth"),
                                          ((256, 256), 4096, "orth"), ((32,), 16, "orth"), ((16, 8, 8), 33, "orth")])
def test_policy_step_value_neglogp(hidden, n, src):
    orc, g = pair(hidden, src)
    rng = np.random.RandomState(5)
    obs = rng.uniform(-2, 2, (n, 18)).astype(np.float32)
    noise = rng.normal(size=(n, 18)).astype(np.float32)
    a, v, nlp = g.step(obs, noise)
    ra, rv, rnlp = orc.step(obs, noise)
    close(a, ra, msg="action"); close(v, rv, msg="value"); close(nlp, rnlp, msg="neglogp")
    close(g.value(obs), rv)
    mu, _ = orc.forward(obs)
    close(g.act_deterministic(obs), mu)


def test_on_device_noise_is_standard_normal():
    _, g = pair((64, 64))
    obs = np.zeros((4096, 18), np.float32)
    a, _, nlp = g.step(obs)                         # no explicit noise -> counter RNG
    mu = g.act_deterministic(obs)
    sigma = np.exp(g.get_tensor("pi/logstd"))
    z = (a - mu) / sigma
    assert abs(float(z.mean())) < 0.02 and float(z.std()) == pytest.approx(1.0, abs=0.02)
    close(nlp, 0.5 * (z.astype(np.float64) ** 2).sum(1) + 18 * 0.9189385175704956 + np.log(sigma).sum(), rtol=2e-4)
    a2, _, _ = g.step(obs)
    assert np.abs(a2 - a).max() > 0.1               # fresh draw per call


@pytest.mark.parametrize("hidden,n,src", [((4, 5), 64, "ginit"), ((4, 5), 2048, "ckpt"), ((64, 64), 64, "orth"),
                                          ((64, 64), 256, "orth"), ((256, 256), 2048, "orth"), ((16, 8, 8), 48, "orth"),
                                          ((64, 64), 75, "orth"), ((256, 256), 1000, "orth"), ((4, 5), 17, "ckpt"),    # ragged: not multiples of 16
                                          ((256, 256), 512, "orth"), ((256, 256), 4096, "orth")])   # one and eight 64-row chunks per split of the weight-gradient kernel
def test_train_step_losses_gradient_and_weights(hidden, n, src):
    orc, g = pair(hidden, src)
    for it in range(3):
        mb = H.synth_minibatch(orc, n, seed=50 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, ref_grad = orc.loss_grad(*args, CR)
        _, ref_norm = orc.clip(ref_grad)
        losses = g.train_step(LR, CR, *args)
        orc.train_step(LR, CR, *args)
        grad, norm = g.last_grad()
        assert 0.02 < ref_losses[4] < 0.98
        close(losses[:4], ref_losses[:4], rtol=1e-4, atol=1e-6, msg="losses it=%d" % it)
        # clipfrac is a COUNT of rows with |ratio - 1| > cliprange: a row whose ratio sits within an ulp of the boundary may fall
        # on either side under any fp32 summation order (the oracle accumulates in double), so one row of n is the resolution
        assert abs(float(losses[4]) - float(ref_losses[4])) <= 1.01 / n, "clipfrac it=%d" % it
        gs = float(np.abs(ref_grad).max())
        close(grad, ref_grad, rtol=2e-4, atol=2e-6 * gs, msg="grad it=%d" % it)
        assert norm == pytest.approx(ref_norm, rel=1e-4)
        close(g.get_flat(0), orc.theta, rtol=1e-4, atol=2e-6, msg="theta it=%d" % it)
        close(g.get_flat(1), orc.m, rtol=2e-4, atol=1e-7 * max(1.0, gs), msg="adam m")
        close(g.get_flat(2), orc.v, rtol=4e-4, atol=1e-10, msg="adam v")
    close(g.beta_powers(), orc.pow, rtol=1e-6)


def test_nonfinite_gradient_poisons_weights_like_the_graph():
    orc, g = pair((4, 5), "ginit")
    mb = H.synth_minibatch(orc, 64, seed=1)
    bad = mb["returns"].copy(); bad[3] = np.inf
    g.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], bad, mb["old_neglogp"], mb["old_values"])
    assert np.isnan(g.get_flat()).all()              # G:24493-24543: NaN scale, not an error


def test_gae_and_advantage_normalisation_kernels():
    _, g = pair((4, 5), "ginit")
    rng = np.random.RandomState(4)
    T, E = 16, 4096
    rew, val = rng.normal(size=(T, E)).astype(np.float32), rng.normal(size=(T, E)).astype(np.float32)
    dones = (rng.uniform(size=(T, E)) < 0.05).astype(np.float32)
    lv, ld = rng.normal(size=E).astype(np.float32), (rng.uniform(size=E) < 0.3).astype(np.float32)
    np.testing.assert_array_equal(g.gae(rew, val, dones, lv, ld, GAMMA, LAM), o.gae(rew, val, dones, lv, ld, GAMMA, LAM))
    got = g.gae(rew[:5, :3], val[:5, :3], dones[:5, :3], lv[:3], ld[:3], 0.9, 1.0)      # ragged small case
    np.testing.assert_array_equal(got, o.gae(rew[:5, :3], val[:5, :3], dones[:5, :3], lv[:3], ld[:3], 0.9, 1.0))
    for Tl, El in ((2048, 1), (300, 7), (131, 64)):             # few environments, long rollouts: the LDS form of the scan (gae_long_kernel), bit for bit
        rw, vl = rng.normal(size=(Tl, El)).astype(np.float32), rng.normal(size=(Tl, El)).astype(np.float32)
        dn = (rng.uniform(size=(Tl, El)) < 0.02).astype(np.float32)
        lv2, ld2 = rng.normal(size=El).astype(np.float32), (rng.uniform(size=El) < 0.3).astype(np.float32)
        np.testing.assert_array_equal(g.gae(rw, vl, dn, lv2, ld2, GAMMA, LAM), o.gae(rw, vl, dn, lv2, ld2, GAMMA, LAM))
    for n in (16, 2048, 1000):
        r, v = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
        close(g.adv_normalize(r, v), o.adv_normalize(r, v), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("E", [1, 7, 64, 4096, 5000, 150000])      # 5000: the reward job's apply pass takes a second batch; 150000: chunks longer than a thread keeps in registers
def test_running_statistics_and_normalisation(E):
    _, g = pair((4, 5), "ginit")
    g.norm_init(E)
    nz = o.Normalizer(E, 18)
    rng = np.random.RandomState(7)
    for it in range(6):
        raw = rng.normal(loc=0.5, scale=2.0, size=(E, 18)).astype(np.float32)
        if it == 3:
            raw[0, 0] = 1e4                                       # clip exercised
        rew = rng.normal(size=E).astype(np.float32)
        dn = (rng.uniform(size=E) < 0.2).astype(np.float32)
        training = it != 4                                        # one frozen step (env_normalize.hpp:76,96)
        nz.training = training
        close(g.norm_obs(raw, training), nz.obs(raw), rtol=2e-5, atol=2e-6, msg="obs it=%d" % it)
        close(g.norm_reward(rew, dn, training), nz.reward(rew, dn), rtol=2e-5, atol=2e-6, msg="rew it=%d" % it)
    m, v, c = g.norm_stats(0)
    close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.obs_rms.var, rtol=1e-5); assert c == nz.obs_rms.count
    m, v, c = g.norm_stats(1)
    close(m, nz.ret_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.ret_rms.var, rtol=1e-5); assert c == nz.ret_rms.count
    st = H.ckpt71_stats()                                         # serialise / deserialise round trip with the fixture
    g.set_norm_stats(0, st["obs_rms"]["mean"], st["obs_rms"]["var"], st["obs_rms"]["count"])
    m, v, c = g.norm_stats(0)
    np.testing.assert_array_equal(m, np.float32(st["obs_rms"]["mean"])); assert c == st["obs_rms"]["count"]


def _rollout_pair(hidden, E, T, seed, src="orth", O=18):
    orc, g = pair(hidden, src, O=O) if O != 18 else pair(hidden, src)
    rng = np.random.RandomState(seed)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, O)
    ro, state, last_v = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    g.norm_init(E)
    g.rollout_alloc(E, T)
    return orc, g, nz, ro, noise


@pytest.mark.parametrize("hidden,E,T", [((4, 5), 1, 64), ((64, 64), 32, 8), ((256, 256), 512, 4)])
def test_collect_on_device_env_matches_oracle_rollout(hidden, E, T):
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 21)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)
    np.testing.assert_array_equal(g.rollout_get("dones"), ro["dones"])
    m, v, c = g.norm_stats(0)
    close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); assert c == nz.obs_rms.count


def test_host_env_rollout_api_matches_device_env_path():
    """The Env-on-host path (reset / act / observe / finish) and the fused device-env path are the same arithmetic."""
    E, T = 8, 6
    orc, g, nz, ro, noise = _rollout_pair((64, 64), E, T, 22)
    raw, _, _ = o.seeded_env_step(1234, 0, E, 0, 18)
    g.rollout_reset(raw)
    for t in range(T):
        acts = g.rollout_act(t, noise[t])
        close(acts, ro["actions"][t], msg="actions t=%d" % t)
        raw, rew, dn = o.seeded_env_step(1234, 0, E, t + 1, 18)
        g.rollout_observe(t, raw, rew, dn)
    g.rollout_finish(GAMMA, LAM)
    for f in ("obs", "values", "neglogp", "rewards", "returns", "dones"):
        close(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)


@pytest.mark.parametrize("hidden,E,T", [((64, 64), 40, 6), ((64, 64), 64, 5), ((16, 8, 8), 50, 5), ((64, 64), 100, 4)])
def test_resident_workgroup_serves_several_row_groups(hidden, E, T, monkeypatch):
    """33..64 environments (100: past the limit, the general path on both sides): the one resident workgroup walks them in groups of 32 rows, on the device env (against the
    oracle's rollout and against the three-kernel path) and behind a host Env (against the general path: copy, statistics
    kernel, both towers, copy back); the statistics are summed in a different order there, so agreement is to rounding."""
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 29)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    dev = {f: g.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones")}
    dev["obs_mean"], dev["obs_var"], _ = g.norm_stats(0)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(dev[f], ro[f], rtol=2e-4, atol=2e-5, msg="device env vs oracle: " + f)
    np.testing.assert_array_equal(dev["dones"], ro["dones"])
    g.close()
    monkeypatch.setenv("PPO_HIP_NO_PERSISTENT_COLLECT", "1")
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 29)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), dev[f], rtol=2e-4, atol=2e-5, msg="three-kernel path vs resident: " + f)
    g.close()
    # host Env: counter RNG (the resident form), same transitions for both forms
    rng = np.random.RandomState(11)
    trans = [(rng.uniform(-1, 1, (E, 18)).astype(np.float32), rng.uniform(-1, 1, E).astype(np.float32), (rng.uniform(size=E) < 0.1).astype(np.float32))
             for _ in range(2 * T + 1)]
    outs = {}
    for form in ("resident", "general"):
        monkeypatch.setenv("PPO_HIP_NO_HOST_FUSED", "1" if form == "general" else "0")
        orc, g = pair(hidden)
        g.norm_init(E); g.rollout_alloc(E, T); g.seed(5)
        g.rollout_reset(trans[0][0])
        got = {}; k = 1
        for it in range(2):
            for t in range(T):
                got["act%d_%d" % (it, t)] = g.rollout_act(t, None)
                g.rollout_observe(t, *trans[k]); k += 1
            g.rollout_finish(GAMMA, LAM)
            for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones"):
                got["%s%d" % (f, it)] = g.rollout_get(f)
            got["obs_mean%d" % it], got["obs_var%d" % it], _ = g.norm_stats(0)
        outs[form] = got
        g.close()
    for key in outs["general"]:
        close(outs["resident"][key], outs["general"][key], rtol=2e-4, atol=2e-5, msg="host Env, resident vs general: " + key)


def test_resident_host_kernel_survives_an_unruly_host(monkeypatch):
    """While the resident rollout kernel waits for the host: another entry point that synchronises the stream (the kernel
    parks itself after its bounded wait, the call goes through, the next act relaunches it), a rollout abandoned half way and
    restarted with ppo_rollout_reset, and a ppo_rollout_finish that comes early.  Same bits as one fused launch per env step
    driven the same way; nothing hangs."""
    E, T = 3, 10
    rng = np.random.RandomState(3)
    trans = [(rng.uniform(-1, 1, (E, 18)).astype(np.float32), rng.uniform(-1, 1, E).astype(np.float32), (rng.uniform(size=E) < 0.2).astype(np.float32))
             for _ in range(40)]
    outs = []
    for resident in (True, False):
        monkeypatch.setenv("PPO_HIP_NO_HOST_RESIDENT", "0" if resident else "1")
        monkeypatch.setenv("PPO_HIP_HOST_POLLS", "2000")                 # park after a few milliseconds
        orc, g = pair((64, 64))
        g.norm_init(E); g.rollout_alloc(E, T); g.seed(7)
        got = {}
        g.rollout_reset(trans[0][0]); k = 1
        for t in range(4):                                               # an abandoned rollout
            got["a%d" % t] = g.rollout_act(t, None)
            g.rollout_observe(t, *trans[k]); k += 1
            if t == 1:
                got["stats_mid"] = g.norm_stats(0)[0]                    # synchronises the stream while the kernel waits for the host
                got["theta_mid"] = g.get_flat(0)
        g.rollout_reset(trans[k][0]); k += 1                             # start over
        for t in range(T):
            got["b%d" % t] = g.rollout_act(t, None)
            g.rollout_observe(t, *trans[k]); k += 1
            if t == 6:
                got["value_mid"] = g.value(trans[0][0])                  # a policy evaluation in between (its own launches)
        g.rollout_finish(GAMMA, LAM)
        for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones"):
            got[f] = g.rollout_get(f)
        got["mean"], got["var"], cnt = g.norm_stats(0); got["cnt"] = np.float64(cnt)
        for t in range(3):                                               # a rollout that is finished early
            got["c%d" % t] = g.rollout_act(t, None)
            g.rollout_observe(t, *trans[k]); k += 1
        g.rollout_finish(GAMMA, LAM)
        got["mean2"], got["var2"], cnt = g.norm_stats(0); got["cnt2"] = np.float64(cnt)
        got["rew_early"] = g.rollout_get("rewards")[:3]
        outs.append(got)
        g.close()
    for key in outs[0]:
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=key)


@pytest.mark.parametrize("hidden,E,T", [((64, 64), 100, 6), ((64, 64), 1024, 8), ((64, 64), 2048, 4), ((16, 8, 8), 333, 5)])
def test_cooperative_persistent_rollout(hidden, E, T, monkeypatch):
    """65..2048 environments on the device env: ceil(E / 32) resident workgroups, one launch per rollout, meeting once per env
    step to combine their chunk moments into the common running statistics.  Against the oracle's rollout (explicit noise),
    against the three-kernel path (to rounding: the statistics are summed in a different chunking), two rollouts in a row,
    and twice with the counter RNG: bit-identical (the combine order is fixed, not arrival order)."""
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 31)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    got = {f: g.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones")}
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(got[f], ro[f], rtol=2e-4, atol=2e-5, msg="vs oracle: " + f)
    np.testing.assert_array_equal(got["dones"], ro["dones"])
    m, v, c = g.norm_stats(0)
    close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.obs_rms.var, rtol=1e-5); assert c == nz.obs_rms.count
    m, v, c = g.norm_stats(1)
    close(v, nz.ret_rms.var, rtol=1e-5); assert c == nz.ret_rms.count
    g.collect_synthetic(1234, GAMMA, LAM, None, step0=T, first=False)          # second rollout, counter RNG
    second = {f: g.rollout_get(f) for f in ("obs", "actions", "values", "rewards", "returns")}
    second["mean"], second["var"], _ = g.norm_stats(0)
    g.close()
    # the same two rollouts again: same bits
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 31)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    g.collect_synthetic(1234, GAMMA, LAM, None, step0=T, first=False)
    again = {f: g.rollout_get(f) for f in ("obs", "actions", "values", "rewards", "returns")}
    again["mean"], again["var"], _ = g.norm_stats(0)
    g.close()
    for k in second:
        np.testing.assert_array_equal(second[k], again[k], err_msg=k)
    # and the one-launch-per-kernel path
    monkeypatch.setenv("PPO_HIP_NO_PERSISTENT_COLLECT", "1")
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 31)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    g.collect_synthetic(1234, GAMMA, LAM, None, step0=T, first=False)
    for f in ("obs", "actions", "values", "rewards", "returns"):
        close(g.rollout_get(f), second[f], rtol=3e-4, atol=3e-5, msg="three-kernel path vs cooperative: " + f)
    g.close()


@pytest.mark.parametrize("hidden,E,T", [((64, 64), 1, 40), ((64, 64), 5, 12), ((64, 64), 32, 7), ((4, 5), 1, 30), ((16, 8, 8), 3, 9)])
def test_host_env_small_batches_three_forms_agree(hidden, E, T, monkeypatch):
    """Env on the host, <= 32 environments (the reference's own setting is ONE): (a) the resident kernel that serves the whole
    rollout from one launch, talking to the host through sequence words in pinned memory, (b) one fused launch per env step,
    (c) the general path (copy, statistics kernel, both towers, copy back).  (a) and (b) run the same statements: every rollout
    field and the statistics must be bit-identical, over two rollouts (state carried over); (c) agrees to rounding.  A host that
    pauses longer than the kernel is willing to poll makes it park itself and be relaunched mid-rollout: same bits.  With ONE environment on the
    reference's shape the resident kernel is narrow_rollout1_kernel<.., HOST> -- three waves, the policy's weights in registers (round 5) --;
    PPO_HIP_NO_ROLLOUT1=1 ("resident_tiles") keeps the 32-row resident workgroup there: same bits again."""
    import time
    rng = np.random.RandomState(7)
    trans = [(rng.uniform(-1, 1, (E, 18)).astype(np.float32), rng.uniform(-1, 1, E).astype(np.float32), (rng.uniform(size=E) < 0.1).astype(np.float32))
             for _ in range(2 * T + 1)]
    outs = {}
    for form in ("resident", "resident_parking", "resident_tiles", "fused", "general"):
        monkeypatch.setenv("PPO_HIP_NO_HOST_RESIDENT", "0" if form.startswith("resident") else "1")
        monkeypatch.setenv("PPO_HIP_NO_ROLLOUT1", "1" if form == "resident_tiles" else "0")
        monkeypatch.setenv("PPO_HIP_NO_HOST_FUSED", "1" if form == "general" else "0")
        monkeypatch.setenv("PPO_HIP_HOST_POLLS", "300" if form == "resident_parking" else "150000")
        orc, g = pair(hidden)
        g.norm_init(E); g.rollout_alloc(E, T); g.seed(99)
        got = {}
        g.rollout_reset(trans[0][0])
        k = 1
        for it in range(2):
            for t in range(T):
                got["act%d_%d" % (it, t)] = g.rollout_act(t, None)
                if form == "resident_parking" and t % 3 == 1:
                    time.sleep(0.02)                                   # longer than 300 polls: the kernel parks, the next act relaunches it
                g.rollout_observe(t, *trans[k]); k += 1
            g.rollout_finish(GAMMA, LAM)
            for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones"):
                got["%s%d" % (f, it)] = g.rollout_get(f)
            for which, nm in ((0, "obs"), (1, "ret")):
                m, v, c = g.norm_stats(which)
                got["%s_mean%d" % (nm, it)], got["%s_var%d" % (nm, it)], got["%s_cnt%d" % (nm, it)] = m, v, np.float64(c)
        outs[form] = got
        kc = g.kernel_counts()
        if form.startswith("resident"):
            one_wave = hidden == (64, 64) and E == 1 and form != "resident_tiles"
            assert (kc["narrow_rollout1_kernel"] > 0) == one_wave and (kc["narrow_rollout_kernel"] > 0) == (not one_wave), (form, kc)
        g.close()
    monkeypatch.delenv("PPO_HIP_NO_ROLLOUT1", raising=False)
    assert np.abs(outs["fused"]["actions1"]).max() > 0
    for key in outs["fused"]:
        np.testing.assert_array_equal(outs["resident"][key], outs["fused"][key], err_msg="resident vs fused: " + key)
        np.testing.assert_array_equal(outs["resident_tiles"][key], outs["fused"][key], err_msg="resident (32-row workgroup) vs fused: " + key)
        np.testing.assert_array_equal(outs["resident_parking"][key], outs["fused"][key], err_msg="parking vs fused: " + key)
        close(outs["general"][key], outs["fused"][key], rtol=2e-4, atol=2e-5, msg="general vs fused: " + key)


@pytest.mark.parametrize("hidden,E,T,nmb,epochs,O", [((4, 5), 1, 256, 4, 2, 18), ((64, 64), 16, 16, 4, 3, 18), ((256, 256), 64, 16, 4, 2, 18),
                                                     ((64, 64), 3, 100, 4, 2, 18),          # M = 75 rows: ragged minibatches
                                                     # minibatches of more than 8192 rows: the epoch's index map + statistics and its gather are two launches there
                                                     # (epoch_prepare_kernel + epoch_gather_kernel; up to 8192 rows epoch_prepare_gather_kernel does both)
                                                     ((64, 64), 1100, 16, 2, 1, 18), ((64, 64), 1100, 16, 2, 1, 36)])
def test_update_phase_matches_oracle(hidden, E, T, nmb, epochs, O):
    orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 23, O=O)
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g.rollout_set(f, ro[f])                      # identical inputs: isolate the update arithmetic
    B = E * T
    rng = np.random.RandomState(99)
    perm = np.arange(B, dtype=np.int32); perms = []
    for _ in range(epochs):
        rng.shuffle(perm); perms.append(perm.copy())
    perms = np.stack(perms)
    ref_rows, ref_mean = orc.update(ro, perms, nmb, LR, CR)
    rows, mean = g.update(LR, CR, epochs, nmb, perms)
    close(rows, ref_rows, rtol=2e-4, atol=2e-6, msg="loss rows")
    close(mean, ref_mean, rtol=2e-4, atol=2e-6, msg="mean losses")
    close(g.get_flat(0), orc.theta, rtol=2e-4, atol=5e-6, msg="theta")
    close(g.beta_powers(), orc.pow, rtol=1e-5)
    # replaying the captured graph on the same rollout continues the optimisation identically
    ref_rows2, _ = orc.update(ro, perms, nmb, LR, CR)
    rows2, _ = g.update(LR, CR, epochs, nmb, perms)
    close(rows2, ref_rows2, rtol=3e-4, atol=3e-6, msg="second update")


@pytest.mark.parametrize("hidden,E,T,explicit_noise", [((64, 64), 1, 64, True), ((64, 64), 1, 33, False), ((64, 64), 7, 16, True), ((64, 64), 32, 9, False),
                                                      ((4, 5), 1, 40, True), ((16, 8, 8), 5, 12, False)])
def test_persistent_rollout_is_bitwise_the_per_step_launches(hidden, E, T, explicit_noise, monkeypatch):
    """n_envs <= 32 on the device env: the whole rollout runs in ONE launch of one persistent workgroup (state in LDS, value
    tower batched afterwards).  Same statements as the one-launch-per-env-step kernel: every rollout field, the running
    statistics and the state carried into the NEXT rollout must be bit-identical, with explicit noise and with the counter RNG."""
    outs = []
    for persistent in (True, False):
        monkeypatch.setenv("PPO_HIP_NO_PERSISTENT_COLLECT", "0" if persistent else "1")
        orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 27)
        got = {}
        for it in range(2):                                  # the second rollout continues from the state the first one left
            g.collect_synthetic(1234, GAMMA, LAM, noise if explicit_noise else None, step0=it * T, first=(it == 0))
            for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones"):
                got["%s%d" % (f, it)] = g.rollout_get(f)
            for which, nm in ((0, "obs"), (1, "ret")):
                m, v, c = g.norm_stats(which)
                got["%s_mean%d" % (nm, it)], got["%s_var%d" % (nm, it)], got["%s_cnt%d" % (nm, it)] = m, v, np.float64(c)
        if explicit_noise:                                   # and against the oracle's rollout
            for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
                close(got[f + "0"], ro[f], rtol=2e-4, atol=2e-5, msg=f)
        outs.append(got)
        g.close()
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)


@pytest.mark.parametrize("O", [18, 36])
@pytest.mark.parametrize("E,T,nmb,epochs", [(16, 16, 4, 3), (64, 64, 32, 1), (3, 100, 4, 2), (16, 16, 1, 1)])
def test_deferred_adam_is_bitwise_the_adam_launch(E, T, nmb, epochs, O, monkeypatch):
    """Reference shapes ([64,64] behind 18 observations, and behind 36 -- the hexapod that observes its velocities, env/hexapod_closed_loop_env.hpp:20,61-72:
    a 64-column observation tile whose first-layer matrix is two 512-piece blocks of the prologue's piece map): inside ppo_update the clip + Adam of step k is applied by the prologue of step k+1's train
    kernel (ping-pong parameter sets, weights written straight into the LDS image).  Same expression, same norm order: loss
    rows, weights, both moments, beta powers, the reported norm and the act model after the update must equal the run with
    an adam_kernel launch per step (PPO_HIP_NO_LAZY_ADAM=1) bit for bit -- including a one-step update (nothing to defer to), ragged minibatches, and a second update
    replayed from the graph.  Both sides run the DEFAULT arithmetic (correctly rounded quotient since round 6); tests/test_other_shapes.py does the same under the
    opt-in 1-ulp quotient."""
    outs = []
    monkeypatch.delenv("PPO_HIP_ADAM_FAST", raising=False)
    for lazy in (True, False):
        monkeypatch.setenv("PPO_HIP_NO_LAZY_ADAM", "0" if lazy else "1")
        orc, g, nz, ro, noise = _rollout_pair((64, 64), E, T, 23, O=O)
        for f in ("obs", "actions", "values", "neglogp", "returns"):
            g.rollout_set(f, ro[f])
        rng = np.random.RandomState(99)
        perms = np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(epochs)])
        got = {}
        k0 = g.kernel_counts()
        for it in range(2):
            got["rows%d" % it], got["mean%d" % it] = g.update(LR, CR, epochs, nmb, perms)
        k1 = g.kernel_counts()                                   # (the compile-time shape ran, both widths; minibatches of <= 64 rows take the resident epoch kernel
        ran = {k: k1[k] - k0[k] for k in ("narrow_train_kernel<static>", "narrow_epoch_kernel")}      #  on the deferred side: tests/test_other_shapes.py compares the two)
        assert ran["narrow_train_kernel<static>"] + ran["narrow_epoch_kernel"] > 0 and (lazy or ran["narrow_epoch_kernel"] == 0), ran
        got["theta"], got["m"], got["v"], got["pow"] = g.get_flat(0), g.get_flat(1), g.get_flat(2), g.beta_powers()
        got["norm"] = np.float32(g.last_grad()[1])
        obs = np.random.RandomState(3).uniform(-2, 2, (40, O)).astype(np.float32)
        got["value"] = g.value(obs); got["mu"] = g.act_deterministic(obs)       # the packed image the act kernels read
        mb = H.synth_minibatch(orc, 64, seed=9)                                  # a plain train step afterwards: set 0 holds the weights
        got["step"] = g.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        got["theta2"] = g.get_flat(0)
        outs.append(got)
        g.close()
    assert np.abs(outs[0]["rows0"]).max() > 0 and np.isfinite(outs[0]["theta"]).all()
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)


def test_update_with_device_permutation_is_a_valid_shuffle():
    """perms=NULL: the keyed bijection must visit every row exactly once per epoch (sum of per-minibatch means of a
    permutation-invariant quantity) and runs must be reproducible for a fixed seed."""
    orc, g, nz, ro, noise = _rollout_pair((64, 64), 16, 16, 24)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    theta0 = g.get_flat()
    rows_a, mean_a = g.update(LR, CR, 2, 4, None, seed=7)
    th_a = g.get_flat()
    assert np.isfinite(rows_a).all() and np.abs(th_a - theta0).max() > 0
    ent0 = float(orc.tensor("pi/logstd").sum()) + 18 * 1.4189385175704956          # analytic entropy of the initial policy
    close(rows_a[:, 2], ent0 + np.zeros(8), rtol=0.01)
    g.set_flat(theta0); g.set_flat(np.zeros_like(theta0), 1); g.set_flat(np.zeros_like(theta0), 2)
    g.set_beta_powers([0.9, 0.999])
    rows_b, _ = g.update(LR, CR, 2, 4, None, seed=7)
    np.testing.assert_array_equal(rows_a, rows_b)                                # bitwise reproducible
    np.testing.assert_array_equal(th_a, g.get_flat())
    g.set_flat(theta0); g.set_flat(np.zeros_like(theta0), 1); g.set_flat(np.zeros_like(theta0), 2)
    g.set_beta_powers([0.9, 0.999])
    rows_c, _ = g.update(LR, CR, 2, 4, None, seed=8)
    assert np.abs(rows_c - rows_a).max() > 0                                     # a different shuffle


def test_full_size_config3_properties():
    """BASELINE config 3 at full size (4096 envs x 16 steps, [256,256], 32 minibatches): size-independent properties.
    approxkl/clipfrac are exactly 0 on the first minibatch (old == current), entropy is analytic, the update is
    deterministic, and returns obey the GAE identity R_t - V_t = delta_t + gamma*lam*nnt*(R_{t+1} - V_{t+1})."""
    import ppo_cpp_amd
    g = ppo_cpp_amd.PPOHip(18, 18, [256, 256])
    g.init_orthogonal(0)
    E, T = 4096, 16
    g.norm_init(E); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, GAMMA, LAM)
    val, ret, rew, dn = (g.rollout_get(f) for f in ("values", "returns", "rewards", "dones"))
    adv = ret - val
    gam, lam = np.float32(GAMMA), np.float32(LAM)
    for t in range(T - 1):
        nnt = 1 - dn[t + 1]
        lhs = adv[t]
        rhs = rew[t] + gam * val[t + 1] * nnt - val[t] + gam * lam * nnt * adv[t + 1]
        close(lhs, rhs, rtol=1e-4, atol=1e-5)
    obs = g.rollout_get("obs")
    assert np.abs(obs).max() <= 10.0 and abs(float(obs.mean())) < 0.05 and float(obs.std()) == pytest.approx(1.0, abs=0.05)
    rows, mean = g.update(LR, CR, 1, 32, None, seed=3)
    assert rows.shape == (32, 5) and np.isfinite(rows).all()
    assert rows[0, 3] == pytest.approx(0.0, abs=1e-9) and rows[0, 4] == 0.0
    close(rows[:, 2], np.full(32, 18 * 1.4189385175704956), rtol=1e-3)
    assert (rows[1:, 3] > 0).all()


@pytest.mark.parametrize("graph_rccl", [None, "0", "1"])
def test_rccl_plumbing_single_rank_communicator(graph_rccl, monkeypatch):
    """ppo_dist_init with world_size 1 exercises the run-time RCCL loading, the communicator and the in-stream
    ncclAllReduce of the gradient buffer (the N-rank code path); results must still match the oracle -- with the capture probe deciding whether the collectives ride in the
    update's graph (None), with the eager sequence forced (PPO_HIP_GRAPH_RCCL=0) and with the capture forced (=1; skipped where the probe says this RCCL cannot be captured)."""
    import ppo_cpp_amd
    if graph_rccl == "1":
        probe = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); probe.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
        capturable = probe.dist_graph_collectives(); probe.close()
        if not capturable:
            pytest.skip("this collective library cannot be stream-captured (the probe of ppo_dist_init): forcing it is an error by design")
    if graph_rccl is not None:
        monkeypatch.setenv("PPO_HIP_GRAPH_RCCL", graph_rccl)
    orc, g = pair((64, 64))
    g.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
    for it in range(2):
        mb = H.synth_minibatch(orc, 128, seed=70 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, _, _ = orc.train_step(LR, CR, *args)
        close(g.train_step(LR, CR, *args), ref_losses, rtol=1e-4, atol=1e-6)
    close(g.get_flat(0), orc.theta, rtol=1e-4, atol=2e-6)
    # the data-parallel forms of the running statistics (two all-reduced batch moments) and of the advantage statistics
    # (all-reduced minibatch sums) must reproduce the single-rank arithmetic when the communicator has one rank
    E, T, nmb, epochs = 16, 16, 4, 2
    orc2, g2, nz, ro, noise = _rollout_pair((64, 64), E, T, 31)
    g2.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
    g2.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g2.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)
    m, v, c = g2.norm_stats(0)
    close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.obs_rms.var, rtol=1e-5); assert c == nz.obs_rms.count
    m, v, c = g2.norm_stats(1)
    close(v, nz.ret_rms.var, rtol=1e-5); assert c == nz.ret_rms.count
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g2.rollout_set(f, ro[f])
    rng = np.random.RandomState(5)
    perm = np.arange(E * T, dtype=np.int32); perms = []
    for _ in range(epochs):
        rng.shuffle(perm); perms.append(perm.copy())
    perms = np.stack(perms)
    ref_rows, _ = orc2.update(ro, perms, nmb, LR, CR)
    print("collectives captured into the update graph:", g2.dist_graph_collectives())
    rows, _ = g2.update(LR, CR, epochs, nmb, perms)             # the collectives ride in the graph when the probe of ppo_dist_init passed, eagerly otherwise
    close(rows, ref_rows, rtol=2e-4, atol=2e-6, msg="loss rows under a communicator")
    close(g2.get_flat(0), orc2.theta, rtol=2e-4, atol=5e-6)
    if graph_rccl is not None:
        assert bool(g2.dist_graph_collectives()) == (graph_rccl == "1")
    g.close(); g2.close()


@pytest.mark.parametrize("hidden", [(64, 64), (256, 256)])
def test_peer_allreduce_single_rank_is_bitwise_the_rccl_path(hidden, monkeypatch):
    """The one-shot peer all-reduce (push into the gather slot, flag, rank-ordered sum + sums of squares) with a one-rank
    communicator: the region is this process's own, so the kernels, the sequence / parity protocol, the probe of
    ppo_dist_peer_attach and the hipGraph capture of the peer kernels all run -- and, one slot being added to nothing, the
    rollout and the update must be BIT-identical to the same run over ncclAllReduce.
    [256,256] (round 5): by default the tiles' finishers push and adam_kernel<.., 2> adds the ranks up and forms the norm from one partial per
    Adam workgroup instead of one per 256-element chunk -- the same gradient bits, a norm that may differ in its last place: that form is held
    to the push / sum form (PPO_HIP_NO_PEER_TILES=1, which stays bitwise the RCCL path) at 1e-5 relative / 1e-6 of each field's largest value."""
    import ppo_cpp_amd
    E, T, nmb, epochs = 16, 16, 4, 2
    outs = []
    forms = [("rccl", None), ("peer", "1")] + ([("peer", "0")] if hidden == (256, 256) else [])
    for form, no_tiles in forms:
        peer = form == "peer"
        if no_tiles is not None:
            monkeypatch.setenv("PPO_HIP_NO_PEER_TILES", no_tiles)
        orc, g, nz, ro, noise = _rollout_pair(hidden, E, T, 31)
        g.dist_init(1, 0, ppo_cpp_amd.PPOHip.dist_unique_id())
        if peer:
            assert g.dist_peer_attach([g.dist_peer_export()]), "probe of the peer path failed"
            assert g.dist_graph_collectives()
        else:
            assert not g.dist_peer_active()
        g.collect_synthetic(1234, GAMMA, LAM, noise)
        got = {f: g.rollout_get(f) for f in ("obs", "values", "returns")}
        got["obs_mean"], got["obs_var"], _ = g.norm_stats(0)
        for f in ("obs", "actions", "values", "neglogp", "returns"):
            g.rollout_set(f, ro[f])
        rng = np.random.RandomState(5)
        perms = np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(epochs)])
        for it in range(2):                                  # second update: graph replay, both parities several times over
            got["rows%d" % it], _ = g.update(LR, CR, epochs, nmb, perms)
        got["theta"], got["m"], got["v"] = g.get_flat(0), g.get_flat(1), g.get_flat(2)
        if peer:
            g.dist_peer_enable(False); assert not g.dist_peer_active()
            g.dist_peer_enable(True)
            got["rows2"], _ = g.update(LR, CR, epochs, nmb, perms)
        outs.append(got)
        g.close()
    monkeypatch.delenv("PPO_HIP_NO_PEER_TILES", raising=False)
    assert np.abs(outs[0]["rows0"]).max() > 0
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    if len(outs) == 3:
        for k in outs[1]:
            np.testing.assert_allclose(outs[2][k], outs[1][k], rtol=1e-5, atol=1e-6 * float(np.abs(outs[1][k]).max()), err_msg="tile push vs push / sum kernels: " + k)
        assert not np.array_equal(outs[2]["theta"], np.zeros_like(outs[2]["theta"]))


@pytest.mark.parametrize("hidden,O,A,n", [((512, 512), 18, 18, 64), ((1024, 1024, 1024), 256, 64, 48), ((1024,), 256, 64, 40), ((1100,), 18, 18, 20),
                                          # regular layout, 64-column wave tiles, widths that are not multiples of 256
                                          # (generic policy head instead of the split-K one), wide obs/action vectors
                                          ((448, 448), 18, 18, 40), ((320, 320), 256, 64, 33), ((448, 320), 18, 18, 16),
                                          # odd observation / action / hidden widths (16-column path), four layers in the regular
                                          # layout, and the maximum depth (eight layers: two-tile layout at width 256)
                                          ((100, 60), 7, 3, 21), ((256, 256, 256, 256), 18, 18, 32), ((256,) * 8, 18, 18, 16)])
def test_wide_networks_use_the_two_tile_layout(hidden, O, A, n):
    """Nets whose per-layer activation tiles exceed 160 KB of LDS ([512,512] and up; BASELINE config 5's shape in fp32)
    run through the same kernels with two ping-pong tiles: act outputs, losses, gradients and Adam must match."""
    orc, g = pair(hidden, "orth", O, A)
    rng = np.random.RandomState(5)
    obs = rng.uniform(-1, 1, (n, O)).astype(np.float32)
    noise = rng.normal(size=(n, A)).astype(np.float32)
    a, v, nlp = g.step(obs, noise)
    ra, rv, rnlp = orc.step(obs, noise)
    close(a, ra, msg="action"); close(v, rv, rtol=2e-4, msg="value"); close(nlp, rnlp, rtol=2e-4, msg="neglogp")
    mb = H.synth_minibatch(orc, n, seed=3)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    ref_losses, ref_grad = orc.loss_grad(*args, CR)
    losses = g.train_step(LR, CR, *args)
    orc.train_step(LR, CR, *args)
    grad, _ = g.last_grad()
    close(losses, ref_losses, rtol=2e-4, atol=1e-6, msg="losses")
    close(grad, ref_grad, rtol=5e-4, atol=3e-6 * float(np.abs(ref_grad).max()), msg="grad")
    close(g.get_flat(0), orc.theta, rtol=2e-4, atol=3e-6, msg="theta")


@pytest.mark.parametrize("tag", ["g45", "g6464", "g256"])
def test_committed_golden_run(tag):
    """The HIP path against the committed vectors of tests/golden/{g45,g6464,g256}_run.npz (the reference's shipped [4,5]
    shape with the graph's initial weights; its real network shape [64,64]; BASELINE configs[2]'s [256,256] -- generated by
    oracle/make_golden_run.py, cross-checked there against float64 autograd): rollout, running statistics,
    first-minibatch gradient and global norm, loss rows, weights and Adam slots after 8 train steps."""
    z, hidden, st, wseed = H.golden_run(tag)
    E, T, nmb, epochs = int(z["E"]), int(z["T"]), int(z["nmb"]), int(z["epochs"])
    lr, cr = float(z["lr"]), float(z["cr"])
    import ppo_cpp_amd
    theta0 = H.golden_weights(o.Oracle(18, 18, list(hidden)), wseed).copy()
    g = ppo_cpp_amd.PPOHip(18, 18, list(hidden)); g.set_flat(theta0)
    g.norm_init(E, float(z["gamma"])); g.rollout_alloc(E, T)
    g.collect_synthetic(int(z["seed"]), float(z["gamma"]), float(z["lam"]), z["noise"])
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), z["ro_" + f], rtol=2e-4, atol=2e-5, msg=f)
    np.testing.assert_array_equal(g.rollout_get("dones"), z["ro_dones"])
    m, v, c = g.norm_stats(0)
    close(m, z["obs_mean"], rtol=1e-5, atol=1e-6); close(v, z["obs_var"], rtol=1e-5); assert c == float(z["obs_count"])
    m, v, c = g.norm_stats(1)
    close(v, z["ret_var"], rtol=1e-5); assert c == float(z["ret_count"])
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g.rollout_set(f, z["ro_" + f])
    rows, mean = g.update(lr, cr, epochs, nmb, z["perms"])
    close(rows, z["loss_rows"], rtol=1e-4, atol=1e-6, msg="loss rows")
    close(mean, z["loss_mean"], rtol=1e-4, atol=1e-6)
    close(g.get_flat(0)[::st], z["theta"], rtol=1e-4, atol=1e-6, msg="weights")
    close(g.get_flat(1)[::st], z["adam_m"], rtol=1e-3, atol=1e-7 * float(np.abs(z["adam_m"]).max()) + 1e-9, msg="adam m")
    close(g.get_flat(2)[::st], z["adam_v"], rtol=1e-3, atol=1e-6 * float(np.abs(z["adam_v"]).max()), msg="adam v")
    if wseed is not None:
        assert np.sqrt(np.sum(g.get_flat(0).astype(np.float64) ** 2)) == pytest.approx(float(z["theta_l2"]), rel=1e-5)
    # first-minibatch gradient through the single-step entry point
    g2 = ppo_cpp_amd.PPOHip(18, 18, list(hidden)); g2.set_flat(theta0)
    B = E * T; M = B // nmb
    flat = {k: np.ascontiguousarray(np.swapaxes(z["ro_" + k], 0, 1)).reshape((B,) + z["ro_" + k].shape[2:]) for k in
            ("obs", "actions", "values", "neglogp", "returns")}
    mb = {}
    for k, val in flat.items():
        sh = np.empty_like(val); sh[z["perms"][0]] = val; mb[k] = sh[:M]
    adv = g2.adv_normalize(mb["returns"], mb["values"])
    losses = g2.train_step(lr, cr, mb["obs"], mb["actions"], adv, mb["returns"], mb["neglogp"], mb["values"])
    close(losses, z["loss_rows"][0], rtol=1e-4, atol=1e-6)
    grad, norm = g2.last_grad()
    close(grad[::st], z["grad0"], rtol=5e-4, atol=3e-6 * float(np.abs(z["grad0"]).max()), msg="gradient")
    assert norm == pytest.approx(float(z["norm0"]), rel=1e-4)
    if wseed is not None:
        assert np.sqrt(np.sum(grad.astype(np.float64) ** 2)) == pytest.approx(float(z["grad0_l2"]), rel=1e-4)


def test_error_paths_and_minimal_sizes():
    """Error convention of the boundary (non-zero status + ppo_last_error) and the smallest legal inputs:
    empty batches are refused, one row steps, two rows train (the reference asserts more than one, ppo2.hpp:402), a
    batch that the minibatch count does not divide is refused, and the handle keeps working after a refused call."""
    import ppo_cpp_amd
    orc, g = pair((64, 64))
    rng = np.random.RandomState(0)
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="positive"):
        g.step(np.zeros((0, 18), np.float32), np.zeros((0, 18), np.float32))
    obs = rng.uniform(-1, 1, (1, 18)).astype(np.float32); noise = rng.normal(size=(1, 18)).astype(np.float32)
    a, v, nlp = g.step(obs, noise); ra, rv, rnlp = orc.step(obs, noise)
    close(a, ra); close(v, rv); close(nlp, rnlp)
    mb = H.synth_minibatch(orc, 2, seed=5)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="more than one row"):
        g.train_step(LR, CR, *[x[:1] for x in args])
    ref_losses, _, _ = orc.train_step(LR, CR, *args)
    close(g.train_step(LR, CR, *args), ref_losses, rtol=1e-4, atol=1e-6)
    close(g.get_flat(0), orc.theta, rtol=1e-4, atol=2e-6)
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="rollout"):
        g.update(LR, CR, 1, 4)
    g.norm_init(3); g.rollout_alloc(3, 5)
    g.collect_synthetic(7, GAMMA, LAM)
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="not divisible"):
        g.update(LR, CR, 1, 4)                                       # 15 rows, 4 minibatches
    rows, mean = g.update(LR, CR, 2, 3, None, seed=1)                # 5-row minibatches: ragged tiles, still fine
    assert rows.shape == (6, 5) and np.isfinite(rows).all() and np.isfinite(g.get_flat()).all()
    import ctypes as C
    seven = np.zeros(7, np.float32)
    rc = g.lib.ppo_set_flat(g.h, 0, seven.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(7))
    assert rc != 0 and b"count" in g.lib.ppo_last_error(g.h)


def test_normaliser_flags_and_return_reset():
    """EnvNormalize's norm_obs / norm_reward constructor flags (env_normalize.hpp:75,95) on the device-resident rollout:
    a switched-off branch passes its data through untouched and never updates its statistics; the other branch is the
    oracle's.  ppo_norm_reset_returns zeroes the discounted-return accumulator and keeps the statistics (:111-116)."""
    E, T = 8, 5
    for norm_obs, norm_rew in ((False, True), (True, False)):
        orc, g, nz, ro, noise = _rollout_pair((64, 64), E, T, 31)
        g.norm_set_flags(norm_obs, norm_rew)
        raws, rews = [], []
        raw, _, _ = o.seeded_env_step(1234, 0, E, 0, 18)
        g.rollout_reset(raw)
        for t in range(T):
            raws.append(raw)
            g.rollout_act(t, noise[t])
            raw, rew, dn = o.seeded_env_step(1234, 0, E, t + 1, 18)
            rews.append(rew)
            g.rollout_observe(t, raw, rew, dn)
        g.rollout_finish(GAMMA, LAM)
        _, _, c_obs = g.norm_stats(0); _, _, c_ret = g.norm_stats(1)
        if not norm_obs:
            np.testing.assert_array_equal(g.rollout_get("obs"), np.stack(raws))          # raw observations reach the policy
            assert c_obs == 1e-6 and c_ret == pytest.approx(E * T, rel=1e-6)
            close(g.rollout_get("rewards"), ro["rewards"], rtol=2e-5, atol=2e-6)         # reward branch unaffected
        else:
            np.testing.assert_array_equal(g.rollout_get("rewards"), np.stack(rews))      # unscaled, unclipped rewards
            assert c_ret == 1e-6 and c_obs == pytest.approx(E * (T + 1), rel=1e-6)
            close(g.rollout_get("obs"), ro["obs"], rtol=2e-5, atol=2e-6)
        g.close()
    _, g = pair((4, 5), "ginit")
    g.norm_init(4)
    nz = o.Normalizer(4, 18)
    r1 = np.float32([1, 2, 3, 4]); d0 = np.zeros(4, np.float32)
    close(g.norm_reward(r1, d0), nz.reward(r1, d0), rtol=2e-5, atol=2e-6)
    g.norm_reset_returns(); nz.ret[:] = 0                                                # EnvNormalize::reset
    close(g.norm_reward(r1, d0), nz.reward(r1, d0), rtol=2e-5, atol=2e-6)
    m, v, c = g.norm_stats(1)
    assert c == nz.ret_rms.count and v[0] == pytest.approx(float(nz.ret_rms.var[0]), rel=1e-5)


def test_seed_reaches_the_action_sampler():
    """ppo_seed (PPO2::seed / --seed): the same seed reproduces the exploration noise, another seed changes it, and the
    draws stay standard normal."""
    orc, g = pair((64, 64))
    obs = np.random.RandomState(0).uniform(-1, 1, (256, 18)).astype(np.float32)
    mu = g.act_deterministic(obs)
    g.seed(11); a1, _, _ = g.step(obs); a1b, _, _ = g.step(obs)
    g.seed(11); a2, _, _ = g.step(obs)
    g.seed(12); a3, _, _ = g.step(obs)
    np.testing.assert_array_equal(a1, a2)
    assert np.abs(a1 - a3).max() > 1e-3 and np.abs(a1 - a1b).max() > 1e-3              # new seed / next call: fresh noise
    sigma = np.exp(orc.tensor("pi/logstd"))
    z = (a3 - mu) / sigma
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05


def test_update_rejects_a_bad_permutation():
    import ppo_cpp_amd
    orc, g, nz, ro, noise = _rollout_pair((64, 64), 4, 8, 33)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    perms = np.stack([np.random.RandomState(i).permutation(32) for i in range(2)]).astype(np.int32)
    before = g.get_flat()
    bad = perms.copy(); bad[1, 3] = bad[1, 4]                                            # duplicate destination
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="not a permutation"):
        g.update(LR, CR, 2, 4, bad)
    bad = perms.copy(); bad[0, 0] = 32                                                   # out of range
    with pytest.raises(ppo_cpp_amd.PPOHipError, match="not a permutation"):
        g.update(LR, CR, 2, 4, bad)
    np.testing.assert_array_equal(g.get_flat(), before)                                  # nothing ran
    rows, _ = g.update(LR, CR, 2, 4, perms)
    assert np.isfinite(rows).all()


def test_hip_path_matches_the_interpreted_reference_graph():
    """The HIP path against what the reference's graph file itself computes (tests/golden/g45_graph_run.npz, produced by
    executing G node by node -- oracle/graph_interp.py), without the hand-written oracle in between: act outputs, the
    five losses, raw gradients, the global norm, and weights / Adam slots / beta powers after three train steps with
    exact-tie rows and both clip branches; then the non-finite-norm poisoning.  North-star tolerance: 1e-4 rel."""
    z = H.graph_run()
    g = hip((4, 5))
    g.set_tensors(H.graph_state(z, "init", "w"))
    a, v, nlp = g.step(z["act/obs"], z["act/noise"])
    close(a, z["act/action"]); close(v, z["act/value"]); close(nlp, z["act/neglogp"])
    close(g.act_deterministic(z["act/obs"]), z["act/det_action"])
    lr, cr = [float(x) for x in z["meta/lr_cr"]]
    names = [n for n, _ in g.tensors]
    def flat(prefix, kind):
        return np.concatenate([np.asarray(z["%s/%s:%s" % (prefix, kind, n)], np.float32).reshape(-1) for n in names])
    for s in range(3):
        p = "train%d" % s
        losses = g.train_step(lr, cr, z[p + "/obs"], z[p + "/actions"], z[p + "/advs"], z[p + "/returns"], z[p + "/old_neglogp"], z[p + "/old_values"])
        close(losses, z[p + "/losses"], rtol=1e-4, atol=1e-6, msg=p)
        grad, norm = g.last_grad()
        gref = flat(p, "grad")
        close(grad, gref, rtol=2e-4, atol=3e-6 * float(np.abs(gref).max()), msg=p + " gradient")
        assert norm == pytest.approx(float(z[p + "/global_norm"]), rel=1e-4)
        close(g.get_flat(0), flat(p, "w"), rtol=1e-4, atol=2e-6, msg=p + " weights")
        close(g.get_flat(1), flat(p, "m"), rtol=2e-4, atol=1e-7, msg=p + " adam m")
        close(g.get_flat(2), flat(p, "v"), rtol=4e-4, atol=1e-9, msg=p + " adam v")
        close(g.beta_powers(), z[p + "/beta_pow"], rtol=1e-6)
    bad = z["train2/advs"].copy(); bad[0] = np.inf
    g.train_step(lr, cr, z["train2/obs"], z["train2/actions"], bad, z["train2/returns"], z["train2/old_neglogp"], z["train2/old_values"])
    assert np.isnan(g.get_flat(0)).all() and z["poison/all_nan"].all()


def test_full_size_config4_single_rank_properties():
    """BASELINE configs[3]'s workload on ONE rank (1024 envs x 64 steps, MLP [64,64], 32 minibatches x 2 epochs: the narrow
    kernels at 2048-row minibatches; the per-rank shard of the 8-GPU layout is covered by tests/test_dp_two_ranks.py).
    Rollout against the oracle; then size-independent properties of the update: first-minibatch ratios are exactly one,
    entropy is the closed form, the shuffle leaves the rollout untouched, and a second update from the same state with the
    same seed is bitwise identical (hipGraph replay == first run)."""
    E, T, nmb, epochs = 1024, 64, 32, 2
    orc, g, nz, ro, noise = _rollout_pair((64, 64), E, T, 41)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)
    np.testing.assert_array_equal(g.rollout_get("dones"), ro["dones"])
    theta0, m0, v0, pw0 = g.get_flat(0), g.get_flat(1), g.get_flat(2), g.beta_powers()
    ret0 = g.rollout_get("returns")
    rows, mean = g.update(LR, CR, epochs, nmb, None, seed=5)
    assert rows.shape == (epochs * nmb, 5) and np.isfinite(rows).all()
    assert rows[0, 3] == pytest.approx(0.0, abs=1e-9) and rows[0, 4] == 0.0 and abs(rows[0, 0]) < 1e-6
    logstd = orc.tensor("pi/logstd")
    assert rows[0, 2] == pytest.approx(float(logstd.sum()) + 18 * 1.4189385175704956, rel=1e-5)
    assert (rows[1:, 3] > 0).all() and rows[nmb:, 1].mean() < rows[:nmb, 1].mean()
    np.testing.assert_array_equal(g.rollout_get("returns"), ret0)
    th1 = g.get_flat(0)
    g.set_flat(theta0); g.set_flat(m0, 1); g.set_flat(v0, 2); g.set_beta_powers(pw0)
    rows2, _ = g.update(LR, CR, epochs, nmb, None, seed=5)
    np.testing.assert_array_equal(rows2, rows); np.testing.assert_array_equal(g.get_flat(0), th1)
    # one explicit-permutation epoch against the oracle at the full size (2048-row minibatches)
    g.set_flat(theta0); g.set_flat(m0, 1); g.set_flat(v0, 2); g.set_beta_powers(pw0)
    perms = np.random.RandomState(1).permutation(E * T).astype(np.int32)[None]
    ref_rows, _ = orc.update(ro, perms, nmb, LR, CR)
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g.rollout_set(f, ro[f])
    rows3, _ = g.update(LR, CR, 1, nmb, perms)
    close(rows3[:, :4], ref_rows[:, :4], rtol=2e-4, atol=2e-6, msg="loss rows")
    assert np.abs(rows3[:, 4] - ref_rows[:, 4]).max() <= 2.01 / (E * T // nmb)          # clipfrac: a count (see test_train_step...)
    close(g.get_flat(0), orc.theta, rtol=2e-4, atol=5e-6)


