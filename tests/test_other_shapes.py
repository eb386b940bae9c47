"""The shape-selected fast kernels on shapes OTHER than 18 observations / 18 actions (VERDICT r3 task 2): the reference's second real
shape is 36 observations (observe_velocities, env/hexapod_closed_loop_env.hpp:20,61-72) with the same 18 actions.  Every case checks
the HIP path against the oracle AND which kernel variant ran (ppo_kernel_counts): a silent fall-back to the round-2 kernels would pass
the numbers and fail the assertion."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import helpers as H
from tests.test_hip_parity import pair, hip, close, CR, LR, GAMMA, LAM

pytestmark = pytest.mark.gpu

# (O, A): 36 / 18 = the reference's velocity-observing hexapod; 50 / 40 -> 64-column tiles on both sides; 20 / 40; 7 / 3 (tiny)
WIDE_SHAPES = [(36, 18), (50, 40), (20, 40), (7, 3)]


def delta(after, before):
    return {k: after[k] - before.get(k, 0) for k in after if after[k] - before.get(k, 0)}


@pytest.mark.parametrize("O,A", WIDE_SHAPES + [(64, 64)])
@pytest.mark.parametrize("n", [2048, 256, 1000, 100])
def test_256x256_train_step_runs_the_fast_pair_on_any_obs_act_width(O, A, n):
    """train8_kernel + weight_grad_assemble_kernel for hidden [256,256] behind any O, A <= 64 and ANY minibatch size (the train
    kernel's grid is padded to whole 64-row chunks): losses, gradient, clipped norm, weights and Adam slots against the oracle over three steps."""
    orc, g = pair((256, 256), O=O, A=A, seed=7)
    k0 = g.kernel_counts()
    for it in range(3):
        mb = H.synth_minibatch(orc, n, seed=70 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, ref_grad = orc.loss_grad(*args, CR)
        _, ref_norm = orc.clip(ref_grad)
        losses = g.train_step(LR, CR, *args)
        orc.train_step(LR, CR, *args)
        grad, norm = g.last_grad()
        close(losses[:4], ref_losses[:4], rtol=1e-4, atol=1e-6, msg="losses it=%d" % it)
        assert abs(float(losses[4]) - float(ref_losses[4])) <= 1.01 / n
        gs = float(np.abs(ref_grad).max())
        # (atol 4e-6 of the largest element: with 64 actions the head gradient has thousands of elements three orders below the largest,
        # formed from 1000-row fp32 sums; the 18-action tests use 2e-6)
        close(grad, ref_grad, rtol=2e-4, atol=4e-6 * gs, msg="grad it=%d" % it)
        assert norm == pytest.approx(ref_norm, rel=1e-4)
        close(g.get_flat(0), orc.theta, rtol=1e-4, atol=2e-6, msg="theta it=%d" % it)
        close(g.get_flat(1), orc.m, rtol=2e-4, atol=2e-7 * max(1.0, gs), msg="adam m")
        close(g.get_flat(2), orc.v, rtol=4e-4, atol=1e-10, msg="adam v")
    assert delta(g.kernel_counts(), k0) == {"train8_kernel": 3, "weight_grad_assemble_kernel": 3}
    g.close()


@pytest.mark.parametrize("O,A", WIDE_SHAPES)
@pytest.mark.parametrize("hidden", [(64, 64), (256, 256)])
def test_step_collect_and_update_on_other_widths(O, A, hidden):
    """policy step, a rollout on the device env, and the minibatch-update phase (explicit permutations) against the oracle"""
    orc, g = pair(hidden, O=O, A=A, seed=9)
    rng = np.random.RandomState(5)
    n = 333
    obs = rng.uniform(-2, 2, (n, O)).astype(np.float32); noise = rng.normal(size=(n, A)).astype(np.float32)
    a, v, nlp = g.step(obs, noise)
    ra, rv, rnlp = orc.step(obs, noise)
    close(a, ra, msg="action"); close(v, rv, msg="value"); close(nlp, rnlp, msg="neglogp")
    E, T, nmb, epochs = 64, 16, 4, 2
    noise = rng.normal(size=(T, E, A)).astype(np.float32)
    nz = o.Normalizer(E, O)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    g.norm_init(E); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)
    np.testing.assert_array_equal(g.rollout_get("dones"), ro["dones"])
    m, var, cnt = g.norm_stats(0)
    close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(var, nz.obs_rms.var, rtol=1e-5); assert cnt == nz.obs_rms.count
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g.rollout_set(f, ro[f])
    perms = np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(epochs)])
    k0 = g.kernel_counts()
    ref_rows, _ = orc.update(ro, perms, nmb, LR, CR)
    rows, _ = g.update(LR, CR, epochs, nmb, perms)
    close(rows, ref_rows, rtol=2e-4, atol=2e-6, msg="loss rows")
    close(g.get_flat(), orc.theta, rtol=2e-4, atol=5e-6, msg="weights")
    d = delta(g.kernel_counts(), k0)
    if hidden == (256, 256):
        assert d.get("train8_kernel") == epochs * nmb and d.get("weight_grad_assemble_kernel") == epochs * nmb and "weight_grad_kernel" not in d
    elif A <= 32:
        assert d.get("narrow_train_kernel<static>") == epochs * nmb, d          # two hidden layers of 64, O <= 64, A <= 32: a compile-time shape
    else:
        # more than 32 actions: the runtime-shape narrow kernels when the LDS image fits (20 / 40 does), the general kernels otherwise
        # (50 / 40: 164 KB of weights + tiles)
        assert d.get("narrow_train_kernel<runtime>") == epochs * nmb or d.get("train_fwd_bwd_kernel") == epochs * nmb, d
    g.close()


@pytest.mark.parametrize("O,A", [(36, 18), (18, 18), (64, 32), (5, 2)])
def test_one_environment_rollout_kernel_on_other_widths(O, A, monkeypatch):
    """narrow_rollout1_kernel (one environment, weights in registers) for any O <= 64, A <= 32: bit-identical to the resident workgroup
    form (PPO_HIP_NO_ROLLOUT1=1) and equal to the oracle's rollout."""
    T = 40
    outs = []
    for r1 in (True, False):
        if r1:
            monkeypatch.delenv("PPO_HIP_NO_ROLLOUT1", raising=False)
        else:
            monkeypatch.setenv("PPO_HIP_NO_ROLLOUT1", "1")
        orc, g = pair((64, 64), O=O, A=A, seed=15)
        g.norm_init(1); g.rollout_alloc(1, T)
        g.seed(5)
        g.collect_synthetic(99, GAMMA, LAM, None)
        g.collect_synthetic(99, GAMMA, LAM, None, step0=T, first=False)          # a second rollout continues from the carried state
        kc = g.kernel_counts()
        outs.append(({f: g.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones")}, g.norm_stats(0), g.norm_stats(1), kc))
        g.close()
    assert outs[0][3]["narrow_rollout1_kernel"] == 2 and outs[1][3]["narrow_rollout1_kernel"] == 0 and outs[1][3]["narrow_rollout_kernel"] == 2
    for f in outs[0][0]:
        np.testing.assert_array_equal(outs[0][0][f], outs[1][0][f], err_msg=f)
    for i in (1, 2):
        for x, y in zip(outs[0][i], outs[1][i]):
            np.testing.assert_array_equal(x, y)
    # ... and against the oracle with the explicit-noise form
    monkeypatch.delenv("PPO_HIP_NO_ROLLOUT1", raising=False)
    orc, g = pair((64, 64), O=O, A=A, seed=15)
    noise = np.random.RandomState(3).normal(size=(T, 1, A)).astype(np.float32)
    nz = o.Normalizer(1, O)
    ro, _, _ = o.collect(orc, nz, 99, T, noise, GAMMA, LAM)
    g.norm_init(1); g.rollout_alloc(1, T)
    g.collect_synthetic(99, GAMMA, LAM, noise)
    assert g.kernel_counts()["narrow_rollout1_kernel"] == 1
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        close(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, msg=f)
    g.close()


@pytest.mark.parametrize("O,E,T,nmb,explicit", [(18, 1, 2048, 32, False), (36, 1, 2048, 32, True), (18, 4, 120, 10, True), (18, 1, 512, 32, False), (36, 2, 24, 1, False),
                                                (18, 1, 7, 7, True), (18, 3, 11, 1, False)])      # one-row minibatches; 33 rows in one minibatch (a second group of ONE row)
def test_resident_epoch_kernel_is_bitwise_the_launch_per_train_step(O, E, T, nmb, explicit, monkeypatch):
    """narrow_epoch_kernel (ppo_narrow.hpp): on the reference's own shape -- [64,64], minibatches of <= 64 rows (ppo2.cpp:114-128: 1 environment x 2048 steps, 32
    minibatches) -- ALL minibatches of an epoch run in one launch whose 2 - 4 workgroups keep the weight image in LDS, the Adam moments in registers and meet once per
    minibatch over their partial gradient vectors -- through one XCD's L2 (the default) or write-through (PPO_HIP_NO_NARROW_EPOCH_XL=1).  Same arithmetic in the
    same order as narrow_train_kernel<.., LAZY> + narrow_reduce_kernel per step (PPO_HIP_NO_NARROW_EPOCH=1): the loss rows of every train step, the last step's gradient and norm, weights, both moments, the beta powers and the next rollout's
    actions / values (the packed image the act kernels read is written back at the kernel's exit) must be the same BITS after two three-epoch updates -- 64-row
    minibatches (two row groups per tower), 48 rows (a ragged second group), 16 rows (one group), one minibatch per epoch; 18 and 36 observations."""
    rng = np.random.RandomState(5)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    perms = [np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(3)]) if explicit else None for _ in range(2)]
    obs2 = rng.uniform(-1, 1, (E, O)).astype(np.float32); nz2 = rng.normal(size=(E, 18)).astype(np.float32)
    outs = []
    for mode in ("one XCD", "write-through", "launches"):
        monkeypatch.setenv("PPO_HIP_NO_NARROW_EPOCH", "1" if mode == "launches" else "0")
        monkeypatch.setenv("PPO_HIP_NO_NARROW_EPOCH_XL", "1" if mode == "write-through" else "0")
        g = hip((64, 64), O=O); g.init_orthogonal(2)
        g.norm_init(E); g.rollout_alloc(E, T)
        g.collect_synthetic(55, GAMMA, LAM, noise)
        k0 = g.kernel_counts()
        acc = []
        for u in range(2):
            rows, mean = g.update(LR, CR, 3, nmb, perms[u], seed=9 + u)
            gr, nrm = g.last_grad()
            acc += [rows.copy(), mean.copy(), gr.copy(), np.float32(nrm), g.get_flat(0), g.get_flat(1), g.get_flat(2), np.asarray(g.beta_powers()).copy()]
            acc += [np.asarray(x).copy() for x in g.step(obs2, nz2)]
        ran = delta(g.kernel_counts(), k0)
        if mode != "launches":
            assert ran.get("narrow_epoch_kernel", 0) >= 3 and "narrow_train_kernel<static>" not in ran, ran    # (one per epoch; a captured graph counts once per capture)
        else:
            assert "narrow_epoch_kernel" not in ran and ran.get("narrow_train_kernel<static>", 0) > 0, ran
        outs.append(acc)
        g.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    outs = [outs[0], outs[2]]
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][2]).max() > 0
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_resident_epoch_kernel_follows_a_changing_minibatch_count(monkeypatch):
    """One handle, three updates with 8, then 32, then 4 minibatches (the XCD-local form's per-step partial buffers grow with the count; the captured graph is rebuilt):
    bit-identical to the launch per train step."""
    E, T = 2, 128
    noise = np.random.RandomState(8).normal(size=(T, E, 18)).astype(np.float32)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("PPO_HIP_NO_NARROW_EPOCH", mode)
        g = hip((64, 64)); g.init_orthogonal(4)
        g.norm_init(E); g.rollout_alloc(E, T)
        g.collect_synthetic(77, GAMMA, LAM, noise)
        acc = []
        for u, nmb in enumerate((8, 32, 4)):
            rows, mean = g.update(LR, CR, 2, nmb, None, seed=20 + u)
            acc += [rows.copy(), mean.copy(), g.get_flat(0), g.get_flat(1)]
        kc = g.kernel_counts()
        assert (kc["narrow_epoch_kernel"] > 0) == (mode == "0")
        outs.append(acc)
        g.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)



@pytest.mark.parametrize("O,E", [(256, 4096), (256, 8192), (256, 150000), (64, 300), (192, 5000), (100, 5000)])
def test_running_statistics_of_wide_observations(O, E):
    """EnvNormalize / RunningStatistics (env/env_normalize.hpp:64-116, common/running_statistics.hpp:26-104) for WIDE observations (BASELINE configs[4]: 256):
    norm_batch_kernel deals the observation job as 64-column groups x row splits (obs_cgroup_job; widths that are multiples of 64) instead of row chunks
    (100 columns: the row-chunk form).  Six batches -- a frozen step, a
    clipped value -- against the oracle's two-pass moments at the tolerances of test_running_statistics_and_normalisation; the count exact; then the same
    six batches again on a fresh handle: same bits (fixed combine order)."""
    def run():
        g = hip((256, 256), O, 18)
        g.norm_init(E)
        nz = o.Normalizer(E, O)
        rng = np.random.RandomState(7)
        for it in range(6):
            raw = rng.normal(loc=0.5, scale=2.0, size=(E, O)).astype(np.float32)
            if it == 3:
                raw[0, 0] = 1e4
            training = it != 4
            nz.training = training
            got = g.norm_obs(raw, training)
            if E <= 8192 or it in (0, 5):
                close(got, nz.obs(raw), rtol=2e-5, atol=2e-6, msg="obs it=%d" % it)
            else:
                nz.obs(raw)
        m, v, c = g.norm_stats(0)
        close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.obs_rms.var, rtol=1e-5); assert c == nz.obs_rms.count
        g.close()
        return m, v
    a = run()
    if E <= 8192:
        b = run()
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("O", [18, 256, 100])
def test_running_statistics_with_a_changing_row_count_on_one_handle(O):
    """ONE handle re-initialised (ppo_norm_init) for 5000, 300, 150 000, 64, 4097, 1, 9000 and 5000 environments in turn, three batches each: the statistics kernel's
    chunking, its arrival counters and (256 columns) its column-group dealing all depend on the row count, and whatever a launch of one shape leaves behind in the
    handle's scratch must not disturb the next shape.  Every normalised batch and the statistics against a fresh oracle normaliser
    (RunningStatistics::update, common/running_statistics.hpp:26-104); the count exact."""
    g = hip((256, 256), O, 18)
    rng = np.random.RandomState(11)
    for it, n in enumerate((5000, 300, 150000, 64, 4097, 1, 9000, 5000)):
        g.norm_init(n)
        nz = o.Normalizer(n, O)
        for b in range(3):
            raw = rng.normal(loc=-0.3, scale=1.5, size=(n, O)).astype(np.float32)
            close(g.norm_obs(raw, True), nz.obs(raw), rtol=3e-5, atol=3e-6, msg="shape %d (%d rows), batch %d" % (it, n, b))
        m, v, c = g.norm_stats(0)
        close(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(v, nz.obs_rms.var, rtol=2e-5); assert c == nz.obs_rms.count
    g.close()


@pytest.mark.parametrize("hidden,O", [((256, 256), 18), ((256, 256), 36), ((64, 64), 18), ((512, 256, 256), 18)])
def test_train_steps_of_changing_row_counts_on_one_handle(hidden, O):
    """ONE handle, train steps of 2048, 1000, 64, 4096, 17, 2048 and 333 rows in turn ([256,256]: the tile / row-split counters and per-row-block slots of the fast pair;
    [64,64]: 1 to 128 row groups of partial vectors; the general family): every step's losses, gradient and norm and the weights against the oracle stepping through the
    same sequence -- whatever a launch of one row count leaves in the handle's workspaces (arrival counters, slots, partial vectors beyond the live rows) must not reach
    the next one."""
    orc, g = pair(hidden, O=O, seed=5)
    for it, n in enumerate((2048, 1000, 64, 4096, 17, 2048, 333)):
        mb = H.synth_minibatch(orc, n, seed=70 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, ref_grad = orc.loss_grad(*args, CR)
        _, ref_norm = orc.clip(ref_grad)
        losses = g.train_step(LR, CR, *args)
        orc.train_step(LR, CR, *args)
        grad, norm = g.last_grad()
        close(losses[:4], ref_losses[:4], rtol=1e-4, atol=1e-6, msg="losses, step %d (%d rows)" % (it, n))
        gs = float(np.abs(ref_grad).max())
        close(grad, ref_grad, rtol=2e-4, atol=2e-6 * gs, msg="gradient, step %d (%d rows)" % (it, n))
        assert norm == pytest.approx(ref_norm, rel=1e-4)
        close(g.get_flat(0), orc.theta, rtol=1e-4, atol=3e-6, msg="weights, step %d" % it)
    g.close()


@pytest.mark.parametrize("hidden", [(64, 64), (256, 256)])
def test_one_handle_through_several_rollout_shapes_equals_fresh_handles(hidden):
    """ONE handle taken through five (environments, steps, minibatches) shapes in turn -- ppo_norm_init + ppo_rollout_alloc again, a collect and two updates each
    ([64,64]: the per-step kernels, the one-wave rollout + the resident epoch kernel, the cooperative rollout, ...; [256,256]: the fast pair at several minibatch sizes)
    -- against a FRESH handle per shape started from the same weights, Adam slots and powers: rollout, loss rows and weights must be the same BITS.  Buffers are
    re-allocated, captured graphs rebuilt and meeting tables carried over between the shapes; nothing of one shape may leak into the next."""
    shapes = ((64, 16, 4), (1, 512, 8), (1024, 64, 32), (1, 2048, 32), (48, 32, 4)) if hidden == (64, 64) else ((64, 16, 4), (512, 16, 8), (4096, 4, 8), (100, 10, 5))
    g = hip(hidden); g.init_orthogonal(7)
    for E, T, nmb in shapes:
        state = (g.get_flat(0), g.get_flat(1), g.get_flat(2), np.asarray(g.beta_powers()).copy())
        outs = []
        for fresh in (False, True):
            h = hip(hidden) if fresh else g
            if fresh:
                h.set_flat(state[0]); h.set_flat(state[1], 1); h.set_flat(state[2], 2); h.set_beta_powers(state[3])
            h.norm_init(E); h.rollout_alloc(E, T)
            h.collect_synthetic(31, GAMMA, LAM, None)
            acc = [h.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns")]
            for u in range(2):
                rows, mean = h.update(LR, CR, 2, nmb, None, seed=3 + u)
                acc += [rows.copy(), h.get_flat(0), h.get_flat(1)]
            outs.append(acc)
            if fresh:
                h.close()
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b, err_msg="shape %s" % ((E, T, nmb),))
    g.close()


@pytest.mark.parametrize("O,E,T,nmb,epochs", [(18, 16, 16, 4, 3), (36, 1, 256, 8, 2), (18, 64, 64, 32, 1), (36, 3, 100, 4, 2)])
def test_exact_adam_in_the_deferred_and_resident_forms(O, E, T, nmb, epochs, monkeypatch):
    """The reference's [64,64] shapes keep their fast forms (Adam deferred into the next train launch; the resident epoch kernel for minibatches of <= 64 rows) and
    compute the quotient m alpha / (sqrt(v) + eps) with the correctly rounded square root and division BY DEFAULT (round 6) -- no deviation from the reference's
    arithmetic: the same BITS as an adam_kernel launch per step (PPO_HIP_NO_LAZY_ADAM=1).  PPO_HIP_ADAM_FAST=1 opts every Adam step of a handle into the hardware's
    1-ulp reciprocal / square root: its fast forms and its launches agree bit for bit too, and differ from the default in last bits only."""
    rng = np.random.RandomState(12)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    outs = {}
    for mode in ("default forms", "default launches", "fast forms", "fast launches"):
        monkeypatch.setenv("PPO_HIP_ADAM_FAST", "1" if mode.startswith("fast") else "0")
        monkeypatch.setenv("PPO_HIP_NO_LAZY_ADAM", "1" if mode.endswith("launches") else "0")
        g = hip((64, 64), O=O); g.init_orthogonal(2)
        g.norm_init(E); g.rollout_alloc(E, T)
        g.collect_synthetic(55, GAMMA, LAM, noise)
        acc = []
        for u in range(2):
            rows, mean = g.update(LR, CR, epochs, nmb, None, seed=9 + u)
            acc += [rows.copy(), g.get_flat(0), g.get_flat(1), g.get_flat(2)]
        kc = g.kernel_counts()
        if mode.endswith("forms"):
            assert kc["narrow_epoch_kernel"] + kc["narrow_train_kernel<static>"] > 0 and (kc["narrow_epoch_kernel"] > 0) == (E * T // nmb <= 64), kc
        outs[mode] = acc
        g.close()
    monkeypatch.delenv("PPO_HIP_ADAM_FAST", raising=False); monkeypatch.delenv("PPO_HIP_NO_LAZY_ADAM", raising=False)
    for a, b in zip(outs["default forms"], outs["default launches"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(outs["fast forms"], outs["fast launches"]):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(outs["default forms"][-3], outs["fast forms"][-3])          # (the weights: the 1-ulp quotient moves some last bits)
    np.testing.assert_allclose(outs["default forms"][-3], outs["fast forms"][-3], rtol=1e-4, atol=1e-6)


# raw device buffers (padding included) compared by the interleaved-handles test's report: persistent state first, then what the last train step / epoch left behind
_IL_STATE = ("theta", "adam_m", "adam_v", "thetaT", "par", "beta_pow", "hyper", "nw_img", "nw_theta1", "nw_m1", "nw_v1", "obs_mean", "obs_var", "nz_ret", "cur_done")
_IL_WORK = ("keys", "gidx", "advstats", "mb_obs", "mb_act", "mb_adv", "mb_ret", "mb_val", "mb_nlp", "x0g", "h_pi_0", "h_vf_0", "h_pi_1", "dmug", "dy_pi_1", "dy_vf_1", "dy_pi_0", "dy_vf_0",
            "slots_pi", "slots_vf", "slabs", "dw2_parts", "nw_partials", "grad", "sumsq", "norm_out", "loss_rows")
_IL_SPECS = (((64, 64), 1, 512, 8, 1), ((256, 256), 64, 16, 4, 2), ((64, 64), 2, 128, 4, 3))
_IL_NAMES = ("obs", "actions", "values", "neglogp", "rewards", "returns", "loss rows", "weights", "adam m", "adam v", "beta powers")


@pytest.mark.parametrize("member", [0, 1, 2])
def test_debug_buffer_reads_every_named_buffer(member):
    """ppo_debug_buffer (include/ppo_hip.h): every name the interleaved-handles report compares can be read after a collect and an update on each of its three shapes, and the
    buffers whose content is known from the public getters hold it: the padded weights contain the dense weights, `hyper` the learning rate and clip range of the last update,
    `beta_pow` the powers, `loss_rows` the rows the update returned.  An unknown name is an error."""
    from ppo_cpp_amd.capi import PPOHipError
    hd, E, T, nmb, sd = _IL_SPECS[member]
    g = hip(hd); g.init_orthogonal(sd); g.norm_init(E); g.rollout_alloc(E, T)
    g.collect_synthetic(40, GAMMA, LAM, None, step0=0, first=True)
    rows, mean = g.update(LR, CR, 2, nmb, None, seed=0)
    got = {k: g.debug_buffer(k) for k in _IL_STATE + _IL_WORK}
    for k in ("theta", "adam_m", "adam_v", "thetaT", "par", "grad", "beta_pow", "hyper", "keys", "gidx", "mb_obs", "loss_rows", "obs_mean", "cur_done"):
        assert got[k].size > 0, k
    for which, k in enumerate(("theta", "adam_m", "adam_v")):
        dense, padded = g.get_flat(which), got[k].view(np.float32)
        assert padded.size > dense.size
        assert np.isin(dense, padded).all(), k                                           # every dense element lies somewhere in the padded buffer
    np.testing.assert_array_equal(got["hyper"].view(np.float32), np.float32([LR, CR]))
    np.testing.assert_array_equal(got["beta_pow"].view(np.float32)[2:4], np.asarray(g.beta_powers(), np.float32))
    np.testing.assert_array_equal(got["loss_rows"].view(np.float32)[:rows.size], rows.ravel())
    assert sorted(got["gidx"][:E * T].tolist()) == list(range(E * T))                  # the last epoch's row map is a permutation of the rollout's rows
    with pytest.raises(PPOHipError, match="no buffer named"):
        g.debug_buffer("no_such_buffer")
    g.close()


def _il_perms(i, it, epochs=2):
    """the explicit permutations of member i's update `it` (the oracle leg of the report: the on-device shuffle needs none)"""
    hd, E, T, nmb, sd = _IL_SPECS[i]
    rng = np.random.RandomState(1000 * i + it)
    return np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(epochs)])


def _il_run(members, iterations=3, debug=False, perms=False, before_update=None, after_update=None):
    """the handles `members` of _IL_SPECS in one process, calls interleaved (every member collects, then every member updates); per member: the public outputs in _IL_NAMES order per
    iteration, and (debug) {stage: {buffer: words}} of the raw device buffers -- read with extra synchronous copies between the calls, which is why the asserted run does without them"""
    hs = {}
    for i in members:
        hd, E, T, nmb, sd = _IL_SPECS[i]
        g = hip(hd); g.init_orthogonal(sd); g.norm_init(E); g.rollout_alloc(E, T)
        hs[i] = g
    out = {i: [] for i in members}
    dbg = {i: {} for i in members}
    for it in range(iterations):
        for i in members:
            hs[i].collect_synthetic(40 + i, GAMMA, LAM, None, step0=it * _IL_SPECS[i][2], first=(it == 0))
            out[i] += [hs[i].rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns")]
            if debug:
                dbg[i]["iteration %d, after the collect" % it] = {k: hs[i].debug_buffer(k) for k in _IL_STATE}
        for i in members:
            if before_update:
                before_update(hs, i, it)
            rows, mean = hs[i].update(LR, CR, 2, _IL_SPECS[i][3], _il_perms(i, it) if perms else None, seed=it)
            if after_update:
                after_update(hs, i, it, dbg)
            out[i] += [rows.copy(), hs[i].get_flat(0), hs[i].get_flat(1), hs[i].get_flat(2), np.asarray(hs[i].beta_powers()).copy()]
            if debug:
                dbg[i]["iteration %d, after the update" % it] = {k: hs[i].debug_buffer(k) for k in _IL_STATE + _IL_WORK}
    for g in hs.values():
        g.close()
    return out, dbg


def _il_first_difference(a, b):
    """(index of the first differing public output, its name) of two members' outputs, or None"""
    for j, (x, y) in enumerate(zip(a, b)):
        if not np.array_equal(x, y):
            return j, "iteration %d: %s" % (j // len(_IL_NAMES), _IL_NAMES[j % len(_IL_NAMES)])
    return None


def _il_oracle_leg(out1, it):
    """the oracle's update `it` of member 1 from the public outputs of a run made with explicit permutations: state = what the run itself reported after update it - 1
    (both runs agree there), rollout = what it collected; returns (loss rows, weights)"""
    hd, E, T, nmb, sd = _IL_SPECS[1]
    n = len(_IL_NAMES)
    orc = o.Oracle(18, 18, list(hd))
    prev = out1[n * (it - 1):n * it]
    orc.theta[:] = prev[7]; orc.m[:] = prev[8]; orc.v[:] = prev[9]; orc.pow[:] = prev[10]
    cur = out1[n * it:n * (it + 1)]
    ro = {"obs": cur[0], "actions": cur[1], "values": cur[2], "neglogp": cur[3], "returns": cur[5]}
    rows, _ = orc.update(ro, _il_perms(1, it), nmb, LR, CR)
    return rows, orc.theta.copy()


def _il_counters(seen):
    def hook(hs, i, it, dbg):
        if i == 1:
            seen.setdefault(it, int(np.count_nonzero(hs[1].debug_buffer("dw2_counters"))))
    return hook


def test_two_handles_interleaved_equal_the_same_handles_run_alone():
    """Three handles in one process, their calls interleaved (A collect, B collect, C collect, A update, B update, ...): the reference's shape with the resident epoch kernel, a
    [256,256] handle and a second narrow shape.  Each must produce exactly what it produces alone -- every meeting table, counter and workspace belongs to its handle -- and one
    ppo_update is a pure function of its inputs (one Session::Run, ppo2/ppo2.hpp:430-468).
    History: at the end of round 5 this failed ONLY behind the rest of the -m gpu suite (the [256,256] handle's replayed updates went wrong by 1e-2).  Round 6 found the cause
    with the oracle leg below: the update's captured graph began with a hipMemsetAsync NODE clearing weight_grad_assemble_kernel's arrival counters, and in a long-lived process
    a replay ran that fill in the middle of the kernels behind it (counters non-zero behind the update; eager launches and a graph without the node were right; the handle run
    ALONE was wrong too one replay later).  The library now captures kernel nodes only (tests/test_race_guards.py::test_update_graph_holds_kernel_nodes_only); this test keeps
    its place at the end of the file, i.e. behind most of the suite, where the old form failed."""
    together = _il_run((0, 1, 2))[0]
    alone = {i: _il_run((i,))[0][i] for i in (0, 1, 2)}
    for i in (0, 1, 2):
        d = _il_first_difference(together[i], alone[i])
        assert d is None, "handle %d %s beside the others differs from the same handle alone, first at %s" % (i, _IL_SPECS[i][:4], d[1])


@pytest.mark.parametrize("members", [(0, 1), (1,)])
def test_replayed_updates_match_the_oracle_beside_other_handles(members):
    """The [256,256] handle of the scenario above through FOUR updates with explicit permutations (three replays of its graph), beside the narrow handle and alone: every
    update's loss rows and weights against oracle.update on the rollout the handle itself collected, starting from the state it reported one update earlier -- which run is
    RIGHT, not only whether two runs agree (at the end of round 5 both were wrong) -- and the arrival counters of weight_grad_assemble_kernel are zero behind every update."""
    seen = {}
    out = _il_run(members, iterations=4, perms=True, after_update=_il_counters(seen))[0][1]
    n = len(_IL_NAMES)
    for it in (1, 2, 3):
        rows, theta = _il_oracle_leg(out, it)
        np.testing.assert_allclose(out[n * it + 6][:, :4], rows[:, :4], rtol=2e-4, atol=3e-6, err_msg="loss rows of update %d" % it)
        np.testing.assert_allclose(out[n * it + 7], theta, rtol=2e-4, atol=5e-6, err_msg="weights after update %d" % it)
    assert [seen[k] for k in sorted(seen)] == [0, 0, 0, 0], "arrival counters left non-zero behind an update: %s" % seen
