"""GPU tests of the bf16 matrix-core path (ppo_config.compute_dtype = PPO_BF16; BASELINE configs[4]: 256 obs / 64 act,
MLP [1024,1024,1024], 8192 envs).  The reference has no bf16 arithmetic, so parity is against the build's own exact-fp32
path / the oracle at a STATED tolerance (~1e-2 relative on the losses): bf16 operands carry 8
significant bits, accumulation, loss arithmetic, gradients' reduction, clip and Adam are fp32."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu

CR = 0.16102319955825806
LR = 0.000393141177482903
GAMMA, LAM = 0.99, 0.95
BF16 = 1


def pair_bf16(hidden, O, A, seed=3):
    import ppo_cpp_amd
    orc = o.Oracle(O, A, list(hidden))
    orc.init_orthogonal(seed)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(seed + 1).uniform(-1.0, 0.2, (1, A))
    g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=BF16)
    g.set_flat(orc.theta)
    return orc, g


def cosine(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.mark.parametrize("hidden,O,A,n", [((1024, 1024, 1024), 256, 64, 256), ((256, 256), 18, 18, 200), ((300, 200), 40, 7, 130),
                                          ((300, 200), 40, 7, 100), ((256, 128), 18, 18, 384),            # 128-row tiles (rows % 256 != 0)
                                          ((1024, 1024, 1024), 256, 64, 1024), ((1024, 512), 300, 100, 1000), ((384,), 18, 18, 640),      # 16 stages per weight-gradient tile: the work-balanced split
                                          ((128, 2048), 18, 18, 256), ((1280,), 18, 70, 128),     # the head kernel's reduction ranges longer than one LDS image: 512 = 2 images; 320 = one and a quarter (8 column blocks)
                                          # configs[4]'s own minibatch (64 stages per tile): ~55 s of scalar fp32 oracle, a soak case (tools/soak_suite.sh); the 4096-row
                                          # properties, chain and assembly cases below keep that size in -m gpu without the oracle
                                          pytest.param((1024, 1024, 1024), 256, 64, 4096, marks=pytest.mark.slow)])
def test_bf16_step_and_train_step_against_fp32_oracle(hidden, O, A, n):
    """act outputs, the five losses, every gradient tensor and one Adam step of the bf16 path against the fp32 oracle.
    Tolerances: values 3e-2, actions / neglogp 5e-3 (the policy head's gain is 0.01), vf_loss 3 % relative, entropy exact
    (it only involves logstd), gradient direction cosine > 0.995 per tensor and global norm within 3 %."""
    orc, g = pair_bf16(hidden, O, A)
    np.testing.assert_array_equal(g.get_flat(0), orc.theta)               # fp32 master weights round-trip untouched
    rng = np.random.RandomState(5)
    obs = rng.uniform(-1, 1, (n, O)).astype(np.float32)
    noise = rng.normal(size=(n, A)).astype(np.float32)
    a, v, nlp = g.step(obs, noise)
    ra, rv, rnlp = orc.step(obs, noise)
    np.testing.assert_allclose(a, ra, rtol=0, atol=5e-3, err_msg="action")
    np.testing.assert_allclose(v, rv, rtol=3e-2, atol=3e-2, err_msg="value")
    np.testing.assert_allclose(nlp, rnlp, rtol=1e-4, atol=5e-3, err_msg="neglogp")
    np.testing.assert_allclose(g.act_deterministic(obs), orc.forward(obs)[0], rtol=0, atol=5e-3)
    np.testing.assert_allclose(g.value(obs), rv, rtol=3e-2, atol=3e-2)
    mb = H.synth_minibatch(orc, n, seed=3)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    ref_losses, ref_grad = orc.loss_grad(*args, CR)
    losses = g.train_step(LR, CR, *args)
    grad, norm = g.last_grad()
    _, ref_norm = orc.clip(ref_grad.copy())
    assert losses[1] == pytest.approx(ref_losses[1], rel=3e-2), "vf_loss"
    assert losses[0] == pytest.approx(ref_losses[0], abs=1e-2), "pg_loss"
    assert losses[2] == pytest.approx(ref_losses[2], rel=1e-5), "entropy"
    assert losses[3] == pytest.approx(ref_losses[3], rel=5e-2, abs=1e-4), "approxkl"
    assert losses[4] == pytest.approx(ref_losses[4], abs=0.03), "clipfrac"
    assert norm == pytest.approx(ref_norm, rel=3e-2)
    for name, off, shape in orc.tensors:
        cnt = int(np.prod(shape))
        gt, rt = grad[off:off + cnt], ref_grad[off:off + cnt]
        if np.linalg.norm(rt) > 1e-3 * ref_norm:                          # tensors that matter to the step
            assert cosine(gt, rt) > 0.995, (name, cosine(gt, rt))
    orc.train_step(LR, CR, *args)
    th = g.get_flat(0)
    # first Adam step moves every weight by ~lr * sign(g): the two paths agree wherever the gradient's sign is resolved
    frac = np.mean(np.abs(th - orc.theta) < 0.2 * LR)
    assert frac > 0.93, frac
    assert np.abs(th - orc.theta).max() <= 2.2 * LR
    np.testing.assert_allclose(g.beta_powers(), orc.pow, rtol=1e-6)


@pytest.mark.parametrize("E,T,nmb", [(64, 8, 4), (100, 10, 5)])
def test_bf16_matches_the_fp32_hip_path_over_an_update(E, T, nmb):
    """Same rollout inputs, same permutations: loss rows of the bf16 path track the library's exact-fp32 path within 1e-2
    (relative on vf_loss / entropy, absolute on the near-zero terms) over 2 epochs at 256/64/[1024]^3: 128-row minibatches (the epoch's observations
    are written once, as bf16 operand rows, by epoch_gather4_kernel) and 200-row ones (not a multiple of the GEMM's row padding: every minibatch is staged by itself)."""
    import ppo_cpp_amd
    hidden, O, A, epochs = (1024, 1024, 1024), 256, 64, 2
    orc, gb = pair_bf16(hidden, O, A)
    gf = ppo_cpp_amd.PPOHip(O, A, list(hidden)); gf.set_flat(orc.theta)
    rng = np.random.RandomState(9)
    noise = rng.normal(size=(T, E, A)).astype(np.float32)
    perms = np.stack([rng.permutation(E * T) for _ in range(epochs)]).astype(np.int32)
    rows = {}
    for name, g in (("f32", gf), ("bf16", gb)):
        g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
        g.collect_synthetic(1234, GAMMA, LAM, noise)
        rows[name] = g.update(LR, CR, epochs, nmb, perms)[0]
    np.testing.assert_allclose(gb.rollout_get("obs"), gf.rollout_get("obs"), rtol=1e-6, atol=1e-6)       # normalisation stays fp32
    np.testing.assert_allclose(gb.rollout_get("values"), gf.rollout_get("values"), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(gb.rollout_get("returns"), gf.rollout_get("returns"), rtol=3e-2, atol=5e-2)
    rb, rf = rows["bf16"], rows["f32"]
    np.testing.assert_allclose(rb[:, 1], rf[:, 1], rtol=2e-2, err_msg="vf_loss")
    np.testing.assert_allclose(rb[:, 2], rf[:, 2], rtol=1e-4, err_msg="entropy")
    np.testing.assert_allclose(rb[:, 0], rf[:, 0], atol=1e-2, err_msg="pg_loss")
    np.testing.assert_allclose(rb[:, 3], rf[:, 3], rtol=1e-2, atol=2e-3, err_msg="approxkl")
    np.testing.assert_allclose(rb[:, 4], rf[:, 4], atol=0.06, err_msg="clipfrac")
    assert cosine(gb.get_flat(0) - orc.theta, gf.get_flat(0) - orc.theta) > 0.9          # the two runs moved the weights the same way


def test_bf16_full_size_config4_properties():
    """BASELINE configs[4] at its per-GPU size (8192 envs x 16 steps, 256 obs / 64 act, MLP [1024,1024,1024], 32 minibatches of
    4096 rows): too large for the oracle, so size-independent properties -- first-epoch ratios are exactly one (act and
    train forward are the same bf16 GEMMs: approxkl = clipfrac = 0, pg_loss ~ 0 because advantages are normalised),
    entropy is the closed form, the loss rows are finite, the value loss falls over the epochs, every rank-local shuffle is
    a permutation (returns unchanged), and a second update from the same state with the same seed is bitwise identical."""
    import ppo_cpp_amd
    hidden, O, A, E, T, nmb, epochs = (1024, 1024, 1024), 256, 64, 8192, 16, 32, 2
    g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=BF16)
    g.init_orthogonal(0)
    g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, GAMMA, LAM, None)
    ret0 = g.rollout_get("returns")
    theta0, m0, v0, pw0 = g.get_flat(0), g.get_flat(1), g.get_flat(2), g.beta_powers()
    rows, mean = g.update(LR, CR, epochs, nmb, None, seed=77)
    assert rows.shape == (epochs * nmb, 5) and np.isfinite(rows).all()
    assert rows[0, 3] == 0.0 and rows[0, 4] == 0.0 and abs(rows[0, 0]) < 1e-5            # ratio == 1 on the first minibatch
    np.testing.assert_allclose(rows[:, 2], 64 * 1.4189385175704956, rtol=1e-3)           # entropy: logstd ~ 0 after a few steps
    assert rows[nmb:, 1].mean() < rows[:nmb, 1].mean()                                    # value loss goes down
    np.testing.assert_array_equal(g.rollout_get("returns"), ret0)
    th1 = g.get_flat(0)
    assert np.isfinite(th1).all() and 0 < np.abs(th1 - theta0).max() <= 2 * epochs * nmb * LR
    g.set_flat(theta0); g.set_flat(m0, 1); g.set_flat(v0, 2); g.set_beta_powers(pw0)
    rows2, _ = g.update(LR, CR, epochs, nmb, None, seed=77)
    np.testing.assert_array_equal(rows2, rows); np.testing.assert_array_equal(g.get_flat(0), th1)


@pytest.mark.parametrize("hidden,O,A,n", [pytest.param((1024, 1024, 1024), 256, 64, 4096, marks=pytest.mark.slow), ((512, 512), 64, 18, 2048), ((1024, 1024, 1024), 256, 64, 1024)])
def test_chained_layers_launch_is_bitwise_the_launch_per_layer(hidden, O, A, n, monkeypatch):
    """gemm_chain_bf16_kernel (ppo_bf16.hpp): the hidden layers of the forward pass, and of the backward pass, as ONE launch each -- a workgroup waits for the
    tiles_j workgroups of its row group only, which share one XCD's L2 (checked in the kernel against the hardware's XCC id).  Same tiles, same main loop,
    same epilogues: losses, gradient, norm, weights and both moments of three train steps must be the same BITS as with PPO_HIP_NO_BF16_CHAIN=1 (a launch per
    layer), at configs[4]'s own 4096-row minibatch (16 row tiles x 8 column tiles x 2 towers = one workgroup per CU), at a [512,512] net with 2048 rows
    (128 workgroups), and at 1024 rows (4 row tiles per tower: fewer row groups than XCDs -- the chain does not apply and the run must simply agree)."""
    outs = []
    for no_chain in ("0", "1"):
        monkeypatch.setenv("PPO_HIP_NO_BF16_CHAIN", no_chain)
        orc, g = pair_bf16(hidden, O, A)
        acc = []
        for it in range(3):
            mb = H.synth_minibatch(orc, n, seed=11 + it)
            args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
            acc.append(np.asarray(g.train_step(LR, CR, *args)).copy())
            gr, nrm = g.last_grad()
            acc += [gr.copy(), np.float32(nrm)]
        acc += [g.get_flat(0), g.get_flat(1), g.get_flat(2)]
        obs = np.random.RandomState(2).uniform(-1, 1, (n, O)).astype(np.float32)
        acc += list(g.step(obs, np.zeros((n, A), np.float32)))
        g.close()
        outs.append(acc)
    monkeypatch.delenv("PPO_HIP_NO_BF16_CHAIN", raising=False)
    assert np.isfinite(outs[0][-4]).all() and np.abs(outs[0][1]).max() > 0
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("hidden,O,A,n", [pytest.param((1024, 1024, 1024), 256, 64, 4096, marks=pytest.mark.slow), ((512, 512), 64, 18, 2048), ((1024, 1024, 1024), 256, 64, 1024),
                                          pytest.param((1024, 1024, 1024, 1024), 256, 64, 1000, marks=pytest.mark.slow), ((64, 64), 18, 18, 512)])
def test_assembly_clip_and_adam_in_one_launch_are_bitwise_the_two_launches(hidden, O, A, n, monkeypatch):
    """bf16_reduce_adam_kernel (ppo_bf16.hpp): the gradient's assembly from the split-K slabs / slots / bias sums and clip + Adam as ONE persistent launch
    whose 256 workgroups meet once for the global norm, the assembled gradient held in registers meanwhile.  Same chunk arithmetic, same partials, same
    order of the norm's sum: losses, gradient, norm, weights, both moments and the bf16 operand copy (read by the next step's forward pass) of four train
    steps -- the clip bites on these inputs -- must be the same BITS as with PPO_HIP_NO_REDUCE_ADAM=1 (bf16_grad_reduce_kernel + adam_kernel), at
    configs[4]'s net (5 rounds of chunks per wave), a [512,512] net (1 round), a four-layer net (7 rounds: the form that requests the Adam slots after the
    meeting) and a net far smaller than the launch (most workgroups hold nothing)."""
    outs = []
    for two in ("0", "1"):
        monkeypatch.setenv("PPO_HIP_NO_REDUCE_ADAM", two)
        orc, g = pair_bf16(hidden, O, A)
        k0 = g.kernel_counts()
        acc = []
        for it in range(4):
            mb = H.synth_minibatch(orc, n, seed=11 + it)
            args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
            acc.append(np.asarray(g.train_step(LR, CR, *args)).copy())
            gr, nrm = g.last_grad()
            acc += [gr.copy(), np.float32(nrm)]
        acc += [g.get_flat(0), g.get_flat(1), g.get_flat(2), np.asarray(g.beta_powers())]
        k1 = g.kernel_counts()
        assert k1["bf16_reduce_adam_kernel"] - k0["bf16_reduce_adam_kernel"] == (4 if two == "0" else 0)
        g.close()
        outs.append(acc)
    monkeypatch.delenv("PPO_HIP_NO_REDUCE_ADAM", raising=False)
    assert np.isfinite(outs[0][-4]).all() and np.abs(outs[0][1]).max() > 0
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_gradient_behind_an_update_is_rebuilt_on_demand(monkeypatch):
    """The fused assembly + Adam launch does not write the assembled gradient (4 of its 41 bytes per parameter); ppo_get_last_grad rebuilds it from the last train
    step's slabs, slot rows and bias rows with bf16_grad_reduce_kernel.  Behind a captured update, a replayed one and a third (the host code of a train step does
    not run on a replay), asked twice, and once with an act call of MORE rows in between (which reallocates the slot rows: the gradient has to be rebuilt before
    that): the same BITS as a handle whose every step wrote its gradient (PPO_HIP_NO_REDUCE_ADAM=1), and a norm that is the gradient's."""
    hidden, O, A, E, T, nmb, epochs = (512, 512), 64, 18, 64, 8, 4, 2
    rng = np.random.RandomState(3)
    noise = rng.normal(size=(T, E, A)).astype(np.float32)
    perms = np.stack([rng.permutation(E * T) for _ in range(epochs)]).astype(np.int32)
    big_obs = rng.normal(size=(1024, O)).astype(np.float32)
    outs = []
    for two in ("0", "1"):
        monkeypatch.setenv("PPO_HIP_NO_REDUCE_ADAM", two)
        orc, g = pair_bf16(hidden, O, A)
        g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
        acc = []
        for it in range(3):
            g.collect_synthetic(100 + it, GAMMA, LAM, noise)
            acc.append(g.update(LR, CR, epochs, nmb, perms)[0].copy())
            if it == 2: g.value(big_obs)                    # 1024 rows > the 512 the workspaces were sized for
            gr, nrm = g.last_grad()
            gr2, nrm2 = g.last_grad()
            np.testing.assert_array_equal(gr, gr2); assert nrm == nrm2
            assert np.sqrt(np.sum(gr.astype(np.float64) ** 2)) == pytest.approx(nrm, rel=1e-5)
            acc += [gr.copy(), np.float32(nrm)]
        acc.append(g.get_flat(0))
        g.close()
        outs.append(acc)
    monkeypatch.delenv("PPO_HIP_NO_REDUCE_ADAM", raising=False)
    assert np.abs(outs[0][1]).max() > 0
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_chained_launches_of_two_row_counts_on_one_handle(monkeypatch):
    """One handle alternating between 4096-row and 2048-row train steps (32 and 16 row groups per chained launch -- as the act path's row count and the minibatch's
    do in a rollout + update): every shape has its own table of workgroup words (a shared table means a slot is different (link, row group) pairs under different
    shapes, and a stale word could then satisfy a wait).  Same BITS as a launch per layer (PPO_HIP_NO_BF16_CHAIN=1) after an irregular sequence of both."""
    seq = [4096] * 6 + [2048] + [4096] * 3 + [2048] * 2 + [4096] * 2 + [2048] + [4096]
    outs = []
    for no_chain in ("0", "1"):
        monkeypatch.setenv("PPO_HIP_NO_BF16_CHAIN", no_chain)
        orc, g = pair_bf16((1024, 1024, 1024), 256, 64)
        mbs = {n: H.synth_minibatch(orc, n, seed=5 + n) for n in (4096, 2048)}
        acc = []
        for n in seq:
            mb = mbs[n]
            acc.append(np.asarray(g.train_step(LR, CR, mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])).copy())
        acc += [g.get_flat(0), g.get_flat(1), g.get_flat(2)]
        g.close()
        outs.append(acc)
    monkeypatch.delenv("PPO_HIP_NO_BF16_CHAIN", raising=False)
    assert np.isfinite(outs[0][-3]).all()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("hidden", [(512, 512, 512), pytest.param((1024, 1024, 1024), marks=pytest.mark.slow)])
def test_bf16_train_steps_of_changing_row_counts_on_one_handle(hidden):
    """ONE bf16 handle, train steps of 4096, 1000, 2048, 300, 4096 and 130 rows in turn (256- and 128-row tiles, chained and per-layer launches, a different
    work-balanced split of the weight-gradient GEMM every time, the persistent assembly + Adam launch behind each): losses, gradient direction and norm of every step
    against the fp32 oracle at the bf16 tolerances.  The oracle takes the SAME weights before every step (the bf16 path's fp32 master weights are copied over), so a
    step is compared on its own: anything stale from the previous shape would show as a wrong gradient, not as drift."""
    orc, g = pair_bf16(hidden, 256, 64)                     # ([1024]^3 = configs[4]'s net: 70 s of scalar oracle, the soak case; [512]^3 takes the same kernels and splits)
    for it, n in enumerate((4096, 1000, 2048, 300, 4096, 130)):
        orc.theta[:] = g.get_flat(0)
        mb = H.synth_minibatch(orc, n, seed=40 + it)
        args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
        ref_losses, ref_grad = orc.loss_grad(*args, CR)
        _, ref_norm = orc.clip(ref_grad.copy())
        losses = g.train_step(LR, CR, *args)
        grad, norm = g.last_grad()
        assert losses[1] == pytest.approx(ref_losses[1], rel=3e-2), ("vf_loss", it, n)
        assert losses[2] == pytest.approx(ref_losses[2], rel=1e-5), ("entropy", it, n)
        assert norm == pytest.approx(ref_norm, rel=3e-2), ("norm", it, n)
        for name, off, shape in orc.tensors:
            cnt = int(np.prod(shape))
            gt, rt = grad[off:off + cnt], ref_grad[off:off + cnt]
            if np.linalg.norm(rt) > 1e-3 * ref_norm:
                assert cosine(gt, rt) > 0.995, (name, it, n, cosine(gt, rt))
    g.close()


def test_a_failed_chained_launch_is_reported_by_the_call_that_produced_the_garbage(capfd):
    """the chained launch's error words (a row group spread over two XCDs, a time-out) used to be read by ppo_update / ppo_train_step only, so inference and
    rollout calls returned garbage as success.  Now every call that chains its layers looks at them: ppo_step repeats its pass with a launch per layer and returns the
    values a handle that never chained returns (same bits: the chained launch is bitwise the per-layer form), the rollout on the device env returns the error; afterwards
    the handle launches layer by layer.  The failure is injected through ppo_debug_raise_chain_error."""
    from ppo_cpp_amd.capi import PPOHipError
    hidden, O, A, n = (1024, 1024, 1024), 256, 64, 4096
    orc, g = pair_bf16(hidden, O, A)
    rng = np.random.RandomState(2)
    obs = rng.uniform(-1, 1, (n, O)).astype(np.float32); noise = rng.normal(size=(n, A)).astype(np.float32)
    a0, v0, nlp0 = g.step(obs, noise)                                # chained, no error
    g.debug_raise_chain_error()
    a1, v1, nlp1 = g.step(obs, noise)                                # reports on stderr, repeats layer by layer
    assert "repeated with a launch per layer" in capfd.readouterr().err
    for x, y in ((a0, a1), (v0, v1), (nlp0, nlp1)):
        np.testing.assert_array_equal(x, y)
    a2, v2, _ = g.step(obs, noise)                                   # the handle stays on the per-layer form: no further report
    assert capfd.readouterr().err == "" and np.array_equal(a2, a0) and np.array_equal(v2, v0)
    with pytest.raises(PPOHipError, match="does not chain"):
        g.debug_raise_chain_error()
    g.close()
    # the rollout path: the error surfaces from ppo_collect_synthetic itself
    orc, g = pair_bf16(hidden, O, A)
    E, T = 4096, 2
    g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, GAMMA, LAM, None)
    g.debug_raise_chain_error()
    with pytest.raises(PPOHipError, match="gemm_chain_bf16_kernel"):
        g.collect_synthetic(1234, GAMMA, LAM, None, step0=T, first=False)
    g.collect_synthetic(1234, GAMMA, LAM, None, step0=T, first=False)      # layer by layer from now on
    assert np.isfinite(g.rollout_get("values")).all()
    g.close()
