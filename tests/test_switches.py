"""Every run-time switch that selects a kernel FORM on one GPU, against the ORACLE (not only against its sibling form): one rollout on the device env with explicit
exploration noise and one two-epoch update with explicit permutations per switch, on a shape that reaches the form the switch selects, at the tolerances of
tests/test_hip_parity.py (bf16 switches: the bf16 path's tolerances against the fp32 oracle).  The data-parallel switches have their cases in
tests/test_dp_two_ranks.py::test_data_parallel_switches_against_the_oracle, the host-Env ones in tests/test_host_layer.py::test_host_env_switches_against_the_oracle;
tests/test_switch_docs.py holds the list of switches to the documentation.  Reference arithmetic: ppo2/runner.hpp:56-191 + ppo2/ppo2.hpp:264-335 as restated in oracle/."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
LR, CR, GAMMA, LAM = 3.93141e-4, 0.161023, 0.99, 0.95

# (switch, hidden, E, T, minibatches): the shape decides which default form the switch replaces
CASES = [("PPO_HIP_NO_GRAPH", (64, 64), 16, 16, 4), ("PPO_HIP_NO_GRAPH", (256, 256), 64, 16, 4),
         ("PPO_HIP_NO_T8", (256, 256), 64, 16, 4), ("PPO_HIP_NO_DW2", (256, 256), 64, 16, 4),
         ("PPO_HIP_NO_LAZY_ADAM", (64, 64), 16, 16, 4), ("PPO_HIP_NO_NARROW", (64, 64), 16, 16, 4),
         ("PPO_HIP_NO_NARROW_EPOCH", (64, 64), 16, 16, 4), ("PPO_HIP_NO_NARROW_EPOCH_XL", (64, 64), 16, 16, 4),
         ("PPO_HIP_ADAM_FAST", (64, 64), 16, 16, 4), ("PPO_HIP_NO_PERSISTENT_COLLECT", (64, 64), 16, 16, 4),
         ("PPO_HIP_NO_ROLLOUT1", (64, 64), 1, 256, 4)]


@pytest.mark.parametrize("switch,hidden,E,T,nmb", CASES)
def test_switch_against_the_oracle(switch, hidden, E, T, nmb, monkeypatch):
    import ppo_cpp_amd
    monkeypatch.setenv(switch, "1")
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(3)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(4).uniform(-1.0, 0.2, (1, 18))
    g = ppo_cpp_amd.PPOHip(18, 18, list(hidden)); g.set_flat(orc.theta)
    rng = np.random.RandomState(5)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    g.norm_init(E); g.rollout_alloc(E, T)
    g.collect_synthetic(1234, GAMMA, LAM, noise)
    for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
        np.testing.assert_allclose(g.rollout_get(f), ro[f], rtol=2e-4, atol=2e-5, err_msg="%s: rollout %s" % (switch, f))
    for f in ("obs", "actions", "values", "neglogp", "returns"):
        g.rollout_set(f, ro[f])                                      # identical inputs: isolate the update arithmetic
    perms = np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(2)])
    ref_rows, _ = orc.update(ro, perms, nmb, LR, CR)
    rows, _ = g.update(LR, CR, 2, nmb, perms)
    np.testing.assert_allclose(rows[:, :4], ref_rows[:, :4], rtol=2e-4, atol=3e-6, err_msg="%s: loss rows" % switch)
    assert np.abs(rows[:, 4] - ref_rows[:, 4]).max() <= 1.01 / (E * T // nmb)
    np.testing.assert_allclose(g.get_flat(0), orc.theta, rtol=2e-4, atol=5e-6, err_msg="%s: weights" % switch)
    np.testing.assert_allclose(g.get_flat(1), orc.m, rtol=3e-4, atol=1e-7, err_msg="%s: Adam m" % switch)
    g.close()


@pytest.mark.parametrize("switch", ["PPO_HIP_NO_BF16_CHAIN", "PPO_HIP_NO_REDUCE_ADAM"])
def test_bf16_switch_against_the_fp32_oracle(switch, monkeypatch):
    """the bf16 path's two form switches at a shape both default forms take (1024-row minibatches of [256,256] behind 64 / 20: four 256-row tiles per tower chain their
    layers; assembly + clip + Adam in one launch): the train step against the fp32 oracle at the bf16 path's tolerances (tests/test_bf16_path.py)"""
    import ppo_cpp_amd
    from tests import helpers as H
    monkeypatch.setenv(switch, "1")
    hidden, O, A, n = (256, 256), 64, 20, 1024
    orc = o.Oracle(O, A, list(hidden)); orc.init_orthogonal(3)
    g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=1); g.set_flat(orc.theta)
    mb = H.synth_minibatch(orc, n, seed=70)
    args = (mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"])
    ref_losses, ref_grad = orc.loss_grad(*args, CR)
    losses = g.train_step(LR, CR, *args)
    grad, norm = g.last_grad()
    np.testing.assert_allclose(losses[1:3], ref_losses[1:3], rtol=2e-2, err_msg="vf_loss, entropy")
    np.testing.assert_allclose(losses[0], ref_losses[0], atol=1e-2, err_msg="pg_loss")
    cos = float(np.dot(grad, ref_grad) / (np.linalg.norm(grad) * np.linalg.norm(ref_grad)))
    assert cos > 0.99, cos
    orc.train_step(LR, CR, *args)
    np.testing.assert_allclose(g.get_flat(0), orc.theta, rtol=0, atol=2.5 * LR)          # an Adam step moves a weight by at most ~lr: the signs agree where the gradient is not noise
    g.close()
