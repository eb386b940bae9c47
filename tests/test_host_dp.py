"""Data parallel in the C++ host layer: `ppo_cpp_hip --ranks N` (ppo_cpp_amd/host/main.cpp, host/ppo2/dist.hpp).

The launcher makes no GPU call; it starts N rank processes of the same binary, serves their control plane (an all-gather over socket
pairs: the ncclUniqueId, the peer exchange's IPC handles, barriers) and returns the worst exit code.  Every rank builds E / N
environments with the global ids rank * E/N + i behind VecEnv + EnvNormalize and runs the unchanged PPO2::learn (reference stack:
ppo2.cpp:188-217, 250; ppo2/ppo2.hpp:239-377).  On the one test GPU the ranks share device 0 and the collective library is the
shared-memory stand-in (tests/fake_rccl), as in tests/test_dp_two_ranks.py; the reference arithmetic is oracle.collect +
oracle.update over the UNION of the ranks' environments, update by update."""
import os
import subprocess

import numpy as np
import pytest

from ppo_cpp_amd import hostapi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, CR, GAMMA, LAM, ENT = 0.000393141177482903, 0.16102319955825806, 0.99, 0.95, 0.0007160293171182275


def driver():
    from ppo_cpp_amd import build as b
    return b.build_driver() if os.path.exists("/opt/rocm/bin/hipcc") else os.path.join(os.path.dirname(hostapi.__file__), "ppo_cpp_hip")


@pytest.mark.parametrize("world", [1, 2, 8])
def test_launcher_control_plane_without_a_gpu(world):
    """The launcher + the ranks' control plane alone (world-size-N processes on CPU): rank 0's 128 bytes reach everybody (the unique
    id's path), a 64-byte all-gather comes back in rank order (the IPC handles'), a barrier; only rank 0's output is relayed."""
    out = subprocess.run([driver(), "--ranks", str(world), "--ctl_selftest"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "ctl_selftest rank 0 of %d: ok" % world


def test_launcher_returns_the_worst_exit_code_when_a_rank_leaves():
    """Failure drill: rank 1 of 4 exits with code 7 while the others wait in a barrier.  The launcher must not hang: it closes the
    control plane (the survivors' calls fail at once), reaps everybody and returns 7."""
    env = dict(os.environ, PPO_CTL_SELFTEST_DIE="1")
    out = subprocess.run([driver(), "--ranks", "4", "--ctl_selftest"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 7, (out.returncode, out.stderr)
    assert "the launcher is gone" in out.stderr


def test_launcher_rejects_environments_that_do_not_divide():
    out = subprocess.run([driver(), "--ranks", "2", "--threads", "3", "--seeded"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "does not divide" in out.stderr


def _fake_rccl(tmp):
    so = os.path.join(tmp, "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-lrt"])
    return so


@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,E,T,nmb,epochs,O,collective", [
    (2, (64, 64), 8, 32, 4, 2, 18, "rccl"), (2, (64, 64), 8, 32, 4, 2, 18, "peer"), (2, (256, 256), 8, 16, 4, 2, 18, "rccl"),
    (8, (64, 64), 16, 32, 4, 2, 18, "rccl"), (8, (256, 256), 16, 16, 4, 2, 18, "peer"),
    # the hexapod's other shape (36 observations, env/hexapod_closed_loop_env.hpp:20) through the same path
    (2, (64, 64), 8, 32, 4, 2, 36, "rccl"), (2, (256, 256), 8, 16, 4, 2, 36, "peer")])
def test_cpp_driver_ranks_match_the_oracle_over_the_union(tmp_path, world, hidden, E, T, nmb, epochs, O, collective):
    """`ppo_cpp_hip --ranks W` on SeededEnvMock x E (E / W per rank) for two updates with EXPLICIT exploration noise and epoch permutations,
    against oracle.collect + oracle.update over the union of the ranks' environments: every update's five mean losses, the final weights and
    both running statistics at the tolerances of test_learn_matches_the_oracle_update_by_update; the replicas' weights, Adam slots and
    statistics bit-identical; the checkpoints the ranks write byte-identical."""
    from oracle import oracle as o
    tmp = str(tmp_path)
    A, U = 18, 2
    El = E // world; Bl = El * T; m = Bl // nmb; M = m * world; B = E * T
    orc = o.Oracle(O, A, list(hidden)); orc.init_orthogonal(3)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(4).uniform(-1.0, 0.2, (1, A))
    theta0 = orc.theta.copy()
    rng = np.random.RandomState(78)
    noise = rng.normal(size=(U, T, E, A)).astype(np.float32)
    perms = np.empty((world, U, epochs, Bl), np.int32); gperms = np.empty((U, epochs, B), np.int32)
    for u in range(U):
        for ep in range(epochs):
            for r in range(world):
                p = rng.permutation(Bl).astype(np.int32)
                perms[r, u, ep] = p
                gperms[u, ep, r * Bl:(r + 1) * Bl] = (p // m) * M + r * m + (p % m)      # global minibatch k = the ranks' k-th local minibatches, in rank order
    theta0.tofile(os.path.join(tmp, "theta.f32")); noise.tofile(os.path.join(tmp, "noise.f32")); perms.tofile(os.path.join(tmp, "perms.i32"))
    env = dict(os.environ, PPO_RCCL_LIBRARY=_fake_rccl(tmp), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [driver(), "--ranks", str(world), "--devices", "0", "--collective", collective, "--threads", str(E), "--batch_steps", str(T),
           "--hidden", ",".join(str(x) for x in hidden), "--epochs", str(epochs), "--minibatches", str(nmb), "--steps", str(U * B),
           "--lr", repr(LR), "--cr", repr(CR), "--ent", repr(ENT), "--seeded", "--obs", str(O), "--explicit_dir", tmp, "--dump_dir", tmp,
           "--saves", "1", "--dir", tmp, "--id", "run", "--replica_saves"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.count(",") == 6]
    assert len(lines) == U                                            # rank 0 alone prints the CSV line of every update
    # ---- the oracle over the union -------------------------------------------------------------------------------------
    nz = o.Normalizer(E, O, gamma=GAMMA)
    state = None
    means = []
    for u in range(U):
        ro, state, _ = o.collect(orc, nz, 1234, T, noise[u], GAMMA, LAM, step0=u * T, state=state)
        _, mean = orc.update(ro, gperms[u], nmb, LR, CR)
        means.append(mean)
    got = []
    for r in range(world):
        base = os.path.join(tmp, "rank%d" % r)
        g = {k: np.fromfile(base + "." + k + ".f32", np.float32) for k in ("losses", "theta", "adam_m", "adam_v", "rms")}
        g["counts"] = np.fromfile(base + ".counts.f64", np.float64); g["dist"] = np.fromfile(base + ".dist.i32", np.int32)
        got.append(g)
    for r, g in enumerate(got):
        assert g["dist"][0] == world and g["dist"][2] == world and g["dist"][3] == (1 if collective == "peer" else 0)
        losses = g["losses"].reshape(U, 5)
        for u in range(U):
            np.testing.assert_allclose(losses[u][:4], means[u][:4], rtol=3e-4, atol=3e-6, err_msg="rank %d: mean losses of update %d" % (r, u))
            assert abs(float(losses[u][4]) - float(means[u][4])) <= 1.01 / M, "clipfrac of update %d" % u
        np.testing.assert_allclose(g["theta"], orc.theta, rtol=2e-4, atol=5e-6, err_msg="weights after %d updates" % U)
        np.testing.assert_allclose(g["rms"][:O], nz.obs_rms.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(g["rms"][O:2 * O], nz.obs_rms.var, rtol=1e-5)
        np.testing.assert_allclose(g["rms"][2 * O], nz.ret_rms.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(g["rms"][2 * O + 1], nz.ret_rms.var, rtol=1e-5)
        assert g["counts"][0] == nz.obs_rms.count and g["counts"][1] == nz.ret_rms.count          # the statistics saw the environments of ALL ranks
    assert np.abs(got[0]["theta"] - theta0).max() > 0
    # the CSV line's losses are rank 0's means (printed with %g)
    np.testing.assert_allclose([float(x) for x in lines[-1].split(",")[1:6]], got[0]["losses"].reshape(U, 5)[-1], rtol=1e-5, atol=1e-7)
    # ---- replicas: same bits, same checkpoint bytes ------------------------------------------------------------------------
    for g in got[1:]:
        for k in ("losses", "theta", "adam_m", "adam_v", "rms", "counts"):
            np.testing.assert_array_equal(got[0][k], g[k], err_msg=k)
    for ext in (".index", ".data-00000-of-00001", ".json"):
        ref = open(os.path.join(tmp, "run.pkl.0" + ext), "rb").read()
        assert len(ref) > 0
        for r in range(1, world):
            assert open(os.path.join(tmp, "run.pkl.rank%d.0%s" % (r, ext)), "rb").read() == ref, "rank %d's checkpoint%s differs" % (r, ext)
    import json
    assert json.loads(ref)["n_envs"] == E                             # the side-car holds the JOB's environment count


@pytest.mark.gpu
def test_cpp_driver_ranks_with_their_own_generators(tmp_path):
    """No explicit inputs: on-device exploration noise keyed by the global row and per-rank epoch shuffles.  4 ranks x 2 environments must
    print finite losses with the initial policy's entropy, and stay bit-identical (the dumps)."""
    tmp = str(tmp_path)
    env = dict(os.environ, PPO_RCCL_LIBRARY=_fake_rccl(tmp), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([driver(), "--ranks", "4", "--devices", "0", "--threads", "8", "--batch_steps", "64", "--hidden", "64,64", "--epochs", "2",
                          "--minibatches", "4", "--steps", str(3 * 8 * 64), "--seeded", "--seed", "5", "--dump_dir", tmp], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.count(",") == 6]
    assert len(lines) == 3
    vals = [float(x) for x in lines[-1].split(",")[:6]]
    assert vals[0] > 0 and np.isfinite(vals).all() and vals[3] == pytest.approx(18 * 1.4189385, rel=0.01)
    th = [np.fromfile(os.path.join(tmp, "rank%d.theta.f32" % r), np.float32) for r in range(4)]
    for t in th[1:]:
        np.testing.assert_array_equal(th[0], t)
