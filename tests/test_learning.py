"""The reference's only validation is that it LEARNS (README.md:22-24 and resources/ppo_cl/*.png: the hexapod's episode reward rising from 0 to 4.5 - 5.5 over a run).
DART and the hexapod are out of scope here, so the same check runs on a small learnable task behind the kept Env interface: TargetEnv (host/env/env_mock.hpp) pays
-mean_j (a_j - (W obs)_j)^2 for a fixed matrix W on SeededEnvMock's observation stream, episodes of 100 steps.  PPO2::learn drives 16 of them behind VecEnv + EnvNormalize for
150 updates of [64,64] on the GPU (the library's own exploration noise and shuffles), and the ORACLE runs the same loop on the CPU (runner.hpp:56-191 + ppo2.hpp:264-335 restated
with numpy draws): both reward curves must rise by a stated margin and end within a stated band of each other.  A slow drift in the update (a stale mirror, a wrong power, a
statistics carry) that no three-update parity test sees would flatten or bend one of the curves."""
import numpy as np
import pytest

from oracle import oracle as o

E, T, HIDDEN, UPDATES, NMB, EPOCHS, LR, CR, GAMMA, LAM, EP_LEN, SEED = 16, 64, (64, 64), 150, 4, 4, 2e-3, 0.2, 0.99, 0.95, 100, 1234
RISE = 0.15                     # mean reward of the last 15 updates over the first 15 (oracle, three draw seeds: -1.48 -> -1.16, -1.44 -> -1.04, -1.47 -> -1.06)
BAND = 0.20                     # |HIP - oracle| of the last-15 means: the two legs draw different exploration noise, and the spread over draws is ~0.12

_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def target_matrix(seed, A, O):
    """TargetEnv's W [A][O] (host/env/env_mock.hpp): 0.5 * sym_unit of the counter hash keyed by the seed"""
    key = _splitmix64(((seed & 0xffffffff) << 32) | 0xffffffff)
    W = np.empty((A, O), np.float32)
    for j in range(A):
        for k in range(O):
            W[j, k] = np.float32(0.5) * (np.float32((_splitmix64(key ^ ((j << 32) | k)) >> 32) >> 8) * np.float32(1.0 / 8388608.0) - np.float32(1.0))
    return W


def oracle_learning_curve(draw_seed, updates=UPDATES):
    """the reference's learn() loop on TargetEnv x E with the oracle's arithmetic; returns the mean un-normalised reward of every update's rollout"""
    O = A = 18
    orc = o.Oracle(O, A, list(HIDDEN)); orc.init_orthogonal(0)
    nz = o.Normalizer(E, O, gamma=GAMMA)
    W = target_matrix(SEED, A, O)
    rng = np.random.RandomState(draw_seed)
    raw, _, _ = o.seeded_env_step(SEED, 0, E, 0, O)            # (TargetEnv's observations are SeededEnvMock's)
    obs, dones, step = nz.obs(raw), np.zeros(E, np.float32), 0
    curve = []
    for _ in range(updates):
        ro = {k: np.empty((T, E) + s, np.float32) for k, s in (("obs", (O,)), ("actions", (A,)), ("values", ()), ("neglogp", ()), ("dones", ()), ("rewards", ()))}
        total = 0.0
        for t in range(T):
            ro["obs"][t] = obs
            a, v, nlp = orc.step(obs, rng.normal(size=(E, A)).astype(np.float32))
            ro["actions"][t], ro["values"][t], ro["neglogp"][t], ro["dones"][t] = a, v, nlp, dones
            rew = -np.mean((a - raw @ W.T) ** 2, axis=1).astype(np.float32)
            step += 1
            raw, _, _ = o.seeded_env_step(SEED, 0, E, step, O)
            dones = np.full(E, 1.0 if step % EP_LEN == 0 else 0.0, np.float32)
            total += float(rew.mean())
            obs = nz.obs(raw)
            ro["rewards"][t] = nz.reward(rew, dones)
        _, last_v = orc.forward(obs)
        ro["returns"] = o.gae(ro["rewards"], ro["values"], ro["dones"], last_v, dones, GAMMA, LAM)
        perm, perms = np.arange(E * T, dtype=np.int32), []
        for _ in range(EPOCHS):                                  # identity per update, shuffled cumulatively per epoch (ppo2.hpp:274-288)
            rng.shuffle(perm); perms.append(perm.copy())
        orc.update(ro, np.stack(perms), NMB, LR, CR)
        curve.append(total / T)
    return np.asarray(curve, np.float32)


_CURVE = {}


def _oracle_curve_cached(draw_seed):
    if draw_seed not in _CURVE:
        _CURVE[draw_seed] = oracle_learning_curve(draw_seed)       # ~20 s of scalar C + numpy; shared by the two loops below
    return _CURVE[draw_seed]


def test_target_matrix_is_bounded_and_fixed():
    W = target_matrix(SEED, 18, 18)
    assert W.shape == (18, 18) and np.abs(W).max() < 0.5 and abs(float(W.mean())) < 0.05
    np.testing.assert_array_equal(W, target_matrix(SEED, 18, 18))
    assert not np.array_equal(W, target_matrix(SEED + 1, 18, 18))


def test_oracle_learns_the_target_task_in_a_short_run():
    """CPU only, 40 updates: the oracle's curve already rises (the full-length comparison with the HIP path is the GPU test below)"""
    c = oracle_learning_curve(5, updates=40)
    assert np.isfinite(c).all() and c[:5].mean() < -1.3 and c[-5:].mean() - c[:5].mean() > 0.03, c


@pytest.mark.gpu
@pytest.mark.parametrize("reference_loop", [False, True])
def test_hip_path_learns_like_the_oracle(reference_loop):
    from ppo_cpp_amd import hostapi
    got = hostapi.learn_curve(E, T, list(HIDDEN), UPDATES, NMB, EPOCHS, LR, CR, gamma=GAMMA, lam=LAM, seed=11, reference_loop=reference_loop)
    hip = got["reward_curve"]
    ref = _oracle_curve_cached(5)
    assert np.isfinite(hip).all() and np.isfinite(got["losses"]).all()
    first_h, last_h, first_r, last_r = hip[:15].mean(), hip[-15:].mean(), ref[:15].mean(), ref[-15:].mean()
    msg = "HIP %.3f -> %.3f, oracle %.3f -> %.3f" % (first_h, last_h, first_r, last_r)
    assert abs(first_h - first_r) < 0.08, msg                   # the same task and the same initial policy
    assert last_h - first_h > RISE and last_r - first_r > RISE, msg
    assert abs(last_h - last_r) < BAND, msg


@pytest.mark.gpu
def test_the_256x256_kernel_pair_learns_the_task_too():
    """The same task through train8_kernel / weight_grad_assemble_kernel ([256,256], the headline's kernels; 64-row minibatches padded to whole chunks, 150 updates = 149 replays
    of the update's graph): no oracle leg (the scalar oracle would need minutes at this width), but the curve must start where the task starts and rise like the [64,64] runs do --
    a kernel pair that trained on stale or half-assembled gradients (round 5's open finding did exactly that after enough replays) flattens it."""
    from ppo_cpp_amd import hostapi
    # (learning rate 3e-4: at 1e-3 this width over-steps on every implementation alike -- approxkl 0.05, clipfrac 0.45, a flat curve from the default kernels, the round-2 kernels,
    #  eager launches and the literal reference loop, measured in round 6 -- a property of the hyper-parameters, not of a kernel)
    got = hostapi.learn_curve(E, T, [256, 256], UPDATES, NMB, EPOCHS, 3e-4, CR, gamma=GAMMA, lam=LAM, seed=11)
    c = got["reward_curve"]
    assert np.isfinite(c).all() and np.isfinite(got["losses"]).all()
    first, last = c[:15].mean(), c[-15:].mean()
    assert -1.6 < first < -1.35, first                          # sigma = 1, mu ~ 0: -(1 + |W obs|^2 / A) ~ -1.47
    assert last - first > 0.12, "reward %.3f -> %.3f" % (first, last)      # (-1.46 -> -1.24 measured)
