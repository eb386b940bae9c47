"""The C++ host layer (ppo_cpp_amd/host): VecEnv threading (the reference's only test, test/vecenv_test.cpp, ported),
host utilities, and -- on the GPU -- PPO2::learn through both the HBM-resident path and the literal reference loop."""
import numpy as np
import pytest

from ppo_cpp_amd import hostapi


@pytest.mark.parametrize("n,workers", [(1, 0), (2, 0), (16, 0), (16, 3), (300, 0), (4096, 0)])
def test_vecenv_step_results(n, workers):
    """reference test/vecenv_test.cpp:51-60 runs N = 1, 2, 16; the pooled VecEnv is also checked at 4096."""
    lib = hostapi.load_host_library()
    assert lib.ppo_host_vecenv_check(n, 5, workers) == 0
    if n <= 16:
        assert lib.ppo_host_vecenv_check(n, 301, workers) == 0          # crosses the every-300th-step done of EnvMock


def test_host_utilities():
    assert hostapi.load_host_library().ppo_host_selftest() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("reference_loop", [False, True])
@pytest.mark.parametrize("hidden,E,T,nmb,epochs,O,A", [((64, 64), 8, 32, 4, 2, 18, 18), ((256, 256), 8, 16, 4, 2, 18, 18),
                                                       # the reference's other real shape: the hexapod that also observes its velocities (36 observations,
                                                       # env/hexapod_closed_loop_env.hpp:20,61-72), with ONE environment like its shipped command line and with 8
                                                       ((64, 64), 1, 64, 4, 2, 36, 18), ((256, 256), 8, 16, 4, 2, 36, 18)])
def test_learn_matches_the_oracle_update_by_update(hidden, E, T, nmb, epochs, O, A, reference_loop):
    _learn_against_the_oracle(hidden, E, T, nmb, epochs, O, A, reference_loop)


# every host-Env run-time switch of the library ("Run-time switches") against the ORACLE, on a shape that reaches the form it replaces:
# <= 32 environments of the narrow net (resident / fused host kernels), ONE environment (the transition also through the BAR; a resident kernel that parks itself
# after 50 polls), a net wider than 64 (the policy kernel publishing the actions itself: off, and forced up to 16 blocks of 16 rows)
@pytest.mark.gpu
@pytest.mark.parametrize("switch,hidden,E,T", [("PPO_HIP_NO_HOST_RESIDENT=1", (64, 64), 8, 32), ("PPO_HIP_NO_HOST_FUSED=1", (64, 64), 8, 32), ("PPO_HIP_NO_VRAM_INBOX=1", (64, 64), 1, 64),
                                               ("PPO_HIP_HOST_POLLS=50", (64, 64), 1, 64), ("PPO_HIP_NO_DIRECT_ACT=1", (256, 256), 8, 16), ("PPO_HIP_DIRECT_ACT_MAX_BLOCKS=16", (256, 256), 128, 4)])
def test_host_env_switches_against_the_oracle(switch, hidden, E, T, monkeypatch):
    k, v = switch.split("=")
    monkeypatch.setenv(k, v)
    _learn_against_the_oracle(hidden, E, T, 4, 2, 18, 18, False)


def _learn_against_the_oracle(hidden, E, T, nmb, epochs, O, A, reference_loop):
    """PPO2::learn end to end against the oracle (reference ppo2/ppo2.hpp:264-349 driving ppo2/runner.hpp:56-191): SeededEnvMock x 8
    behind VecEnv + EnvNormalize, two updates with EXPLICIT exploration noise and epoch permutations, through the HBM-resident
    loop and through the literal reference loop (Runner::run, host-side row permutation and slicing, _train_step per minibatch).
    Every update's five mean losses, the final weights and both running statistics must match oracle.collect + oracle.update at
    the fp32 tolerances of test_update_phase_matches_oracle: the done-view bookkeeping, the carry of observations / dones /
    discounted returns from one rollout into the next, the env-major flatten and the per-update permutation are all on this path."""
    from oracle import oracle as o
    LR, CR, GAMMA, LAM = 0.000393141177482903, 0.16102319955825806, 0.99, 0.95
    U, B = 2, E * T
    orc = o.Oracle(O, A, list(hidden)); orc.init_orthogonal(3)
    orc.tensor("pi/logstd")[:] = np.random.RandomState(4).uniform(-1.0, 0.2, (1, A))
    theta0 = orc.theta.copy()
    rng = np.random.RandomState(77)
    noise = rng.normal(size=(U, T, E, A)).astype(np.float32)
    perms = np.empty((U, epochs, B), np.int32)
    for u in range(U):
        perm = np.arange(B, dtype=np.int32)                        # identity per update, shuffled cumulatively per epoch (ppo2.hpp:274-288)
        for e in range(epochs):
            rng.shuffle(perm); perms[u, e] = perm
    got = hostapi.learn_explicit(E, T, list(hidden), theta0, noise, perms, nmb, lr=LR, cliprange=CR, gamma=GAMMA, lam=LAM, reference_loop=reference_loop,
                                 obs_dim=O, act_dim=A)
    nz = o.Normalizer(E, O, gamma=GAMMA)
    state = None
    for u in range(U):
        ro, state, _ = o.collect(orc, nz, 1234, T, noise[u], GAMMA, LAM, step0=u * T, state=state)
        _, mean = orc.update(ro, perms[u], nmb, LR, CR)
        np.testing.assert_allclose(got["losses"][u][:4], mean[:4], rtol=3e-4, atol=3e-6, err_msg="mean losses of update %d" % u)
        assert abs(float(got["losses"][u][4]) - float(mean[4])) <= 1.01 / (B // nmb), "clipfrac of update %d" % u
    np.testing.assert_allclose(got["theta"], orc.theta, rtol=2e-4, atol=5e-6, err_msg="weights after %d updates" % U)
    np.testing.assert_allclose(got["obs_mean"], nz.obs_rms.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(got["obs_var"], nz.obs_rms.var, rtol=1e-5)
    np.testing.assert_allclose(got["ret_mean"], nz.ret_rms.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(got["ret_var"], nz.ret_rms.var, rtol=1e-5)
    assert got["obs_count"][0] == nz.obs_rms.count and got["ret_count"][0] == nz.ret_rms.count


@pytest.mark.gpu
def test_learn_resident_and_reference_loop_agree_on_first_update():
    """Same env stack, same weights, own generators (on-device noise and shuffle vs host shuffle): the two loops differ only in the
    draws, so the entropy (draw independent to first order) agrees tightly.  (The oracle comparison is the test above.)"""
    a = hostapi.learn(8, 32, [64, 64], n_updates=1, nminibatches=4, noptepochs=2)
    b = hostapi.learn(8, 32, [64, 64], n_updates=1, nminibatches=4, noptepochs=2, reference_loop=True)
    assert np.isfinite(a["losses"]).all() and np.isfinite(b["losses"]).all()
    assert a["losses"][2] == pytest.approx(b["losses"][2], rel=1e-3)
    assert a["fps_last"] > 0 and b["fps_last"] > 0


@pytest.mark.gpu
def test_learn_with_reference_envmock_stub():
    """EnvMock (the reference's constant-data stub, one env) drives the whole stack like ppo2.cpp:188-250."""
    r = hostapi.learn(1, 256, [4, 5], n_updates=2, nminibatches=4, noptepochs=2, seeded_env=False)
    assert np.isfinite(r["losses"]).all()
    assert r["losses"][2] == pytest.approx(18 * 1.4189385175704956, rel=1e-2)


@pytest.mark.gpu
def test_command_line_driver_trains_saves_and_plays_back(tmp_path):
    """ppo_cpp_hip with the reference's flags (ppo2.cpp:93-128): 4 updates with 2 saves -> fps CSV lines like
    ppo2.hpp:343-349, checkpoints <id>.pkl.0 / .1 in the reference's format; then playback (--path) of the last one."""
    import os
    import subprocess
    from ppo_cpp_amd import build as b
    exe = b.build_driver() if os.path.exists("/opt/rocm/bin/hipcc") else os.path.join(os.path.dirname(hostapi.__file__), "ppo_cpp_hip")
    out = subprocess.run([exe, "--steps", "4096", "--batch_steps", "256", "--threads", "4", "--hidden", "64,64", "--epochs", "2", "--minibatches", "4",
                          "--lr", "3e-4", "--cr", "0.2", "--saves", "2", "--dir", str(tmp_path), "--id", "run", "--seeded"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.count(",") == 6]
    assert len(lines) == 4                                         # 4096 / (4 envs * 256 steps) updates
    fps, pg, vf, ent, kl, cf = [float(x) for x in lines[-1].split(",")[:6]]
    assert fps > 0 and np.isfinite([pg, vf, ent, kl, cf]).all() and ent == pytest.approx(18 * 1.4189385, rel=0.01)
    for i in (0, 1):
        for ext in (".index", ".data-00000-of-00001", ".json"):
            assert os.path.exists(str(tmp_path / ("run.pkl.%d%s" % (i, ext))))
    play = subprocess.run([exe, "--path", str(tmp_path / "run.pkl.1"), "--hidden", "64,64", "--seeded"], capture_output=True, text=True, timeout=120)
    assert play.returncode == 0, play.stderr
    assert play.stdout.count("action[0..3]") == 5
    # the hexapod's other shape from the command line: 36 observations (observe_velocities), ONE environment like the reference's shipped run
    out = subprocess.run([exe, "--steps", "1024", "--batch_steps", "256", "--threads", "1", "--hidden", "64,64", "--epochs", "2", "--minibatches", "4",
                          "--seeded", "--obs", "36"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.count(",") == 6]
    assert len(lines) == 4 and np.isfinite([float(x) for x in lines[-1].split(",")[:6]]).all()
    assert subprocess.run([exe, "--obs", "36"], capture_output=True, text=True, timeout=60).returncode == 1        # (EnvMock is 18 / 18)


@pytest.mark.gpu
def test_learn_honours_the_normaliser_flags():
    """EnvNormalize(norm_obs=false) / (norm_reward=false) on the HBM-resident learn() path (env_normalize.hpp:75,95):
    the switched-off statistics are never updated, the other ones see every env step."""
    E, T = 8, 16
    a = hostapi.learn(E, T, [64, 64], n_updates=2, nminibatches=4, noptepochs=2, norm_obs=False)
    assert a["obs_count"] == 1e-6 and a["ret_count"] == pytest.approx(2 * E * T, rel=1e-6) and np.isfinite(a["losses"]).all()
    b = hostapi.learn(E, T, [64, 64], n_updates=2, nminibatches=4, noptepochs=2, norm_reward=False)
    assert b["ret_count"] == 1e-6 and b["obs_count"] == pytest.approx(E * (2 * T + 1), rel=1e-6) and np.isfinite(b["losses"]).all()
    c = hostapi.learn(E, T, [64, 64], n_updates=2, nminibatches=4, noptepochs=2, seed=5)
    d = hostapi.learn(E, T, [64, 64], n_updates=2, nminibatches=4, noptepochs=2, seed=5)
    e = hostapi.learn(E, T, [64, 64], n_updates=2, nminibatches=4, noptepochs=2, seed=6)
    assert c["losses"] == d["losses"] and c["losses"] != e["losses"]                    # PPO2::seed drives noise + shuffles


def test_pooled_vecenv_is_clean_under_thread_sanitizer(tmp_path):
    """SURVEY section 5 (race detection): the pooled VecEnv (worker pool, generation counter, chunked env ranges) stepped
    from the main thread under -fsanitize=thread -- construction, 300 steps over 16 environments with 3 workers, reset,
    destruction -- with the reference test's expectations on the rows (test/vecenv_test.cpp:36-47); then 200 environments on 4 threads with
    pauses of 0 / 50 / 400 us between the steps, so that helpers are caught watching the claim word, timing out into their sleep and being
    woken from it; once with the default spin budget and once with PPO_VECENV_SPIN_US=0 (helpers always sleep)."""
    import os
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppo_cpp_amd", "host")
    src = r"""
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <unistd.h>
#include "env/env_mock.hpp"
#include "env/vec_env.hpp"
int main() {
    {
        std::vector<std::shared_ptr<Env>> envs;
        for (int i = 0; i < 200; ++i) envs.push_back(std::make_shared<EnvMock>(i + 1));
        VecEnv ve{envs, 4};
        Mat actions = Mat::Zero(ve.get_num_envs(), ve.get_action_space_size());
        for (int s = 0; s < 120; ++s) {
            const std::vector<Mat> r = ve.step(actions);
            for (int e = 0; e < 200; ++e) if (r[1](e, 0) != (float)(e + 1) || r[0](e, 17) != (float)(e + 1)) return 6;
            if (s % 7 == 3) usleep(400); else if (s % 5 == 1) usleep(50);
        }
        if (ve.pool_workers() != 4 || ve.pool_chunk() < 1) return 7;
    }
    for (int workers : {0, 1, 3}) {
        std::vector<std::shared_ptr<Env>> envs;
        for (int i = 0; i < 16; ++i) envs.push_back(std::make_shared<EnvMock>(i + 1));
        VecEnv ve{envs, workers};
        Mat first = ve.reset();
        if (first.rows() != 16) return 2;
        for (int s = 0; s < 300; ++s) {
            Mat actions = Mat::Zero(ve.get_num_envs(), ve.get_action_space_size());
            const std::vector<Mat> r = ve.step(actions);
            for (int e = 0; e < 16; ++e) {
                if (r[1](e, 0) != (float)(e + 1)) return 3;
                for (int j = 0; j < 18; ++j) if (r[0](e, j) != (float)(e + 1)) return 4;
                if (r[2](e, 0) != ((s + 1) % 300 == 0 ? 1.f : 0.f)) return 5;
            }
        }
    }
    std::puts("ok");
    return 0;
}
"""
    cpp = tmp_path / "tsan_vecenv.cpp"; cpp.write_text(src)
    exe = tmp_path / "tsan_vecenv"
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", host, "-o", str(exe), str(cpp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    for spin in (None, "0"):
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
        env.pop("PPO_VECENV_SPIN_US", None)
        if spin is not None:
            env["PPO_VECENV_SPIN_US"] = spin
        run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
        assert run.returncode == 0 and "ok" in run.stdout and "ThreadSanitizer" not in run.stderr, run.stdout[-500:] + run.stderr[-3000:]
