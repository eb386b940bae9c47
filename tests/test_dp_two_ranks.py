"""Data-parallel path with 2, 4 and 8 ranks on one GPU.

RCCL refuses two ranks on the same device, so the collective library is swapped (PPO_RCCL_LIBRARY) for
tests/fake_rccl: the same five nccl* entry points over POSIX shared memory.  Everything else is the product path: two
processes, one ppo_handle each, ppo_dist_init, the all-reduced running statistics during the rollout, the all-reduced
advantage moments, the gradient all-reduce between the reduce and the Adam kernels.  The reference arithmetic is the
single-process oracle over the UNION of the ranks' environments / minibatch rows."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H  # noqa: F401
from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, CR, GAMMA, LAM = 3.93141e-4, 0.161023, 0.99, 0.95


def close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol, err_msg=msg)


def build_fake_rccl(tmp):
    so = os.path.join(tmp, "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-lrt"])
    return so


def run_workers(tmp, world, fin, env, timeout=900):
    """world processes of tests/dp_worker.py on the one test GPU; returns their output files' contents in rank order"""
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), fin, os.path.join(tmp, "out%d.npz" % r)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("data-parallel workers timed out")
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return [np.load(os.path.join(tmp, "out%d.npz" % r)) for r in range(world)]


SHAPES_UNION = [(2, (64, 64), 16, 16, 4, 2), (2, (256, 256), 64, 8, 4, 1),
                # BASELINE configs[3] at its per-rank workload (1024 envs over 8 GPUs = 128 envs x 64 steps
                # per rank, MLP [64,64], 32 minibatches of 256 rows per rank)
                (2, (64, 64), 256, 64, 32, 1),
                # world 4 and 8 (the target is 8 GPUs): the 8-slot rank-ordered sum, the (rank + k) % world push order, the statistics
                # table of 8 ranks, the regions' capacity at 8 ranks -- one narrow and one [256,256] shape each
                (4, (64, 64), 16, 16, 4, 2), (4, (256, 256), 64, 8, 4, 1),
                (8, (64, 64), 16, 16, 4, 2), (8, (256, 256), 64, 8, 4, 1),
                # ... and configs[3]'s literal strong-scaling shard at 8 ranks: 1024 envs -> 128 per rank, 256 minibatch rows per rank
                (8, (64, 64), 1024, 64, 32, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,E,T,nmb,epochs", SHAPES_UNION)
@pytest.mark.parametrize("peer", [False, True])
def test_two_ranks_equal_the_single_process_oracle_on_the_union(tmp_path, world, hidden, E, T, nmb, epochs, peer):
    """`world` ranks (2, 4, 8) as processes on the one test GPU against the single-process oracle over the union of their rows.
    peer = True: every collective goes through the one-shot peer all-reduce (ppo_peer.hpp) -- the processes map each
    other's gather region over hipIpc (same mechanism as the GPUs of a node; here all regions live on the one test GPU) and
    the whole update, collectives included, replays from the hipGraph."""
    _union_against_the_oracle(tmp_path, world, hidden, E, T, nmb, epochs, peer)


# every data-parallel run-time switch of the library ("Run-time switches") against the ORACLE over the union, not only against its sibling form
@pytest.mark.gpu
@pytest.mark.parametrize("hidden,E,T,nmb", [((256, 256), 64, 8, 4), ((64, 64), 16, 16, 4)])
@pytest.mark.parametrize("peer,switch", [(True, "PPO_HIP_NO_ADAM_MEET=1"), (True, "PPO_HIP_NO_PEER_TILES=1"), (True, "PPO_HIP_PEER_STATS=0"), (True, "PPO_HIP_PEER_TIMEOUT_MS=20000"),
                                         # (PPO_HIP_GRAPH_RCCL=1 FORCES the capture: the shared-memory stand-in cannot be captured, so that value has its case with the
                                         #  real library at world 1, tests/test_hip_parity.py::test_rccl_plumbing_single_rank_communicator)
                                         (True, "PPO_HIP_PEER_REDUCE=0"), (False, "PPO_HIP_GRAPH_RCCL=0")])
def test_data_parallel_switches_against_the_oracle(tmp_path, hidden, E, T, nmb, peer, switch):
    k, v = switch.split("=")
    _union_against_the_oracle(tmp_path, 2, hidden, E, T, nmb, 1, peer, {k: v})


def _union_against_the_oracle(tmp_path, world, hidden, E, T, nmb, epochs, peer, extra_env=None):
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(11)
    theta0 = orc.theta.copy()
    rng = np.random.RandomState(41)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    # local shuffles per rank and the equivalent global one: global minibatch k = rank 0's k-th minibatch rows, then rank 1's
    El = E // world; Bl = El * T; m = Bl // nmb; M = m * world
    perms = np.empty((world, epochs, Bl), np.int32); gperms = np.empty((epochs, E * T), np.int32)
    for ep in range(epochs):
        for r in range(world):
            p = rng.permutation(Bl).astype(np.int32)
            perms[r, ep] = p
            gperms[ep, r * Bl:(r + 1) * Bl] = (p // m) * M + r * m + (p % m)
    uid = np.zeros(128, np.uint8)
    name = ("/ppo_dp_test_%d_%d" % (os.getpid(), rng.randint(1 << 30))).encode()
    uid[:len(name)] = np.frombuffer(name, np.uint8)
    fin = os.path.join(tmp, "in.npz")
    np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=theta0, uid=uid, gamma=GAMMA, lam=LAM, seed=1234,
             noise=noise, perms=perms, lr=LR, cr=CR, **{"ref_" + k: ro[k] for k in ("obs", "actions", "values", "neglogp", "returns")})
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="1" if peer else "0")
    env.update(extra_env or {})
    outs = run_workers(tmp, world, fin, env)
    for out in outs:                                                           # the communicator itself reports `world` ranks
        assert int(out["comm_nranks"]) == world
    # ---- rollout: every rank's shard equals the oracle's columns; the running statistics are over ALL environments ----
    for r, out in enumerate(outs):
        sl = slice(r * El, (r + 1) * El)
        for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
            close(out["ro_" + f], ro[f][:, sl], rtol=2e-4, atol=2e-5, msg="rank %d %s" % (r, f))
        np.testing.assert_array_equal(out["ro_dones"], ro["dones"][:, sl])
        close(out["obs_mean"], nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(out["obs_var"], nz.obs_rms.var, rtol=1e-5)
        close(out["ret_var"], nz.ret_rms.var, rtol=1e-5)
        assert float(out["obs_count"]) == nz.obs_rms.count and float(out["ret_count"]) == nz.ret_rms.count
    for k in ("obs_mean", "obs_var", "ret_mean", "ret_var"):
        for out in outs[1:]:
            np.testing.assert_array_equal(outs[0][k], out[k])                 # bit-identical on every rank
    # ---- update: loss rows and weights equal the oracle's update over the union; replicas stay bit-identical ----------
    ref_rows, _ = orc.update(ro, gperms, nmb, LR, CR)
    for out in outs:
        close(out["rows"], ref_rows, rtol=2e-4, atol=2e-6, msg="loss rows")
        close(out["theta"], orc.theta, rtol=2e-4, atol=5e-6, msg="weights")
    for k in ("rows", "theta", "adam_m", "adam_v"):
        for out in outs[1:]:
            np.testing.assert_array_equal(outs[0][k], out[k])
    assert np.abs(outs[0]["theta"] - theta0).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,E,T,nmb,epochs", [(2, (64, 64), 16, 16, 4, 2), (2, (256, 256), 64, 8, 4, 1), (4, (64, 64), 16, 16, 4, 2)])
def test_two_ranks_literal_global_shuffle(tmp_path, world, hidden, E, T, nmb, epochs):
    """ppo_dist_global_shuffle(1): the reference's sampling taken literally under data parallelism (ppo2/ppo2.hpp:288-307) --
    ONE permutation of the rows of all ranks per epoch after an all-gather of the rollout, rank r training rows [r M, (r+1) M) of every
    global minibatch.  Both ranks get the SAME global permutation the single-process oracle uses over the union: no equivalent-
    permutation construction, the comparison is direct.  Loss rows and weights must match the oracle; the replicas must stay
    bit-identical."""
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(12)
    theta0 = orc.theta.copy()
    rng = np.random.RandomState(43)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    El = E // world; Bl = El * T
    gperms = np.stack([rng.permutation(E * T).astype(np.int32) for _ in range(epochs)])
    perms = np.zeros((world, epochs, Bl), np.int32)                      # unused in this mode
    uid = np.zeros(128, np.uint8)
    name = ("/ppo_dp_gs_%d_%d" % (os.getpid(), rng.randint(1 << 30))).encode()
    uid[:len(name)] = np.frombuffer(name, np.uint8)
    fin = os.path.join(tmp, "in.npz")
    np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=theta0, uid=uid, gamma=GAMMA, lam=LAM, seed=1234,
             noise=noise, perms=perms, gperms=gperms, lr=LR, cr=CR, **{"ref_" + k: ro[k] for k in ("obs", "actions", "values", "neglogp", "returns")})
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="0", PPO_TEST_GLOBAL="1")
    outs = run_workers(tmp, world, fin, env)
    ref_rows, _ = orc.update(ro, gperms, nmb, LR, CR)
    for out in outs:
        close(out["rows"], ref_rows, rtol=2e-4, atol=2e-6, msg="loss rows")
        close(out["theta"], orc.theta, rtol=2e-4, atol=5e-6, msg="weights")
    for k in ("rows", "theta", "adam_m", "adam_v"):
        for out in outs[1:]:
            np.testing.assert_array_equal(outs[0][k], out[k])
    assert np.abs(outs[0]["theta"] - theta0).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,E,T,nmb,epochs", [(2, (256, 256), 64, 8, 4, 1), (2, (384, 256), 128, 16, 4, 2), (8, (256, 256), 128, 8, 4, 1), (2, (256, 256, 256), 64, 8, 4, 1),
                                                         (8, (256, 128, 256), 128, 8, 4, 1)])
@pytest.mark.parametrize("bucketed", [True, False])
def test_two_ranks_bf16_path(tmp_path, world, hidden, E, T, nmb, epochs, bucketed):
    """The bf16 matrix-core path under data parallelism (the configuration SURVEY 8e expects DP to pay for): gradient assembly per rank,
    all-reduce, recomputed sums of squares, fold, clip + Adam.  Against the fp32 oracle over the union at the bf16 path's stated
    tolerances (tests/test_bf16_path.py); the replicas must stay bit-identical.  bucketed (the default, ppo_dist_bucketed): the gradient leaves layer by layer,
    last layer + heads first, each bucket's all-reduce on a second stream under the remaining backward / weight-gradient launches (2 and 3 buckets here);
    False: one all-reduce of the whole vector behind them."""
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(13)
    theta0 = orc.theta.copy()
    rng = np.random.RandomState(47)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    El = E // world; Bl = El * T; m = Bl // nmb; M = m * world
    perms = np.empty((world, epochs, Bl), np.int32); gperms = np.empty((epochs, E * T), np.int32)
    for ep in range(epochs):
        for r in range(world):
            p = rng.permutation(Bl).astype(np.int32)
            perms[r, ep] = p
            gperms[ep, r * Bl:(r + 1) * Bl] = (p // m) * M + r * m + (p % m)
    uid = np.zeros(128, np.uint8)
    name = ("/ppo_dp_bf_%d_%d" % (os.getpid(), rng.randint(1 << 30))).encode()
    uid[:len(name)] = np.frombuffer(name, np.uint8)
    fin = os.path.join(tmp, "in.npz")
    np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=theta0, uid=uid, gamma=GAMMA, lam=LAM, seed=1234,
             noise=noise, perms=perms, lr=LR, cr=CR, **{"ref_" + k: ro[k] for k in ("obs", "actions", "values", "neglogp", "returns")})
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="0", PPO_TEST_BF16="1", PPO_TEST_BUCKETED="1" if bucketed else "0")
    outs = run_workers(tmp, world, fin, env)
    ref_rows, _ = orc.update(ro, gperms, nmb, LR, CR)
    for out in outs:
        rows = out["rows"]
        np.testing.assert_allclose(rows[:, 1], ref_rows[:, 1], rtol=3e-2, err_msg="vf_loss")
        np.testing.assert_allclose(rows[:, 2], ref_rows[:, 2], rtol=1e-4, err_msg="entropy")
        np.testing.assert_allclose(rows[:, 0], ref_rows[:, 0], atol=1e-2, err_msg="pg_loss")
        np.testing.assert_allclose(rows[:, 3], ref_rows[:, 3], rtol=5e-2, atol=2e-3, err_msg="approxkl")
        np.testing.assert_allclose(rows[:, 4], ref_rows[:, 4], atol=0.08, err_msg="clipfrac")
        da, db = (out["theta"] - theta0).astype(np.float64), (orc.theta - theta0).astype(np.float64)
        assert float(da @ db / (np.linalg.norm(da) * np.linalg.norm(db))) > 0.9           # the run moved the weights the oracle's way
        assert np.abs(out["theta"] - orc.theta).max() <= 2.5 * LR * epochs * nmb
    for k in ("rows", "theta", "adam_m", "adam_v"):
        for out in outs[1:]:
            np.testing.assert_array_equal(outs[0][k], out[k])


@pytest.mark.gpu
@pytest.mark.parametrize("world,hidden,E,T", [(2, (64, 64), 16, 12), (8, (256, 256), 64, 6)])
def test_statistics_exchange_inside_the_statistics_kernels_is_bitwise_the_all_reduce_form(tmp_path, world, hidden, E, T):
    """Over peer-mapped regions norm_batch_kernel's last workgroups write this rank's batch moments straight into every rank's gather
    area and norm_finalize_kernel waits for the flags (no push / sum launches between them).  The table that arrives is the table the
    general all-reduce delivered (PPO_HIP_PEER_STATS=0), so every rollout field and the running statistics must be the same BITS."""
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    nmb, epochs = 2, 1
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(14)
    rng = np.random.RandomState(51)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, 18)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    Bl = (E // world) * T
    perms = np.stack([np.stack([rng.permutation(Bl).astype(np.int32) for _ in range(epochs)]) for _ in range(world)])
    res = []
    for mode in ("1", "0"):
        sub = os.path.join(tmp, "m" + mode); os.makedirs(sub)
        uid = np.zeros(128, np.uint8)
        name = ("/ppo_dp_st_%d_%d_%s" % (os.getpid(), rng.randint(1 << 30), mode)).encode()
        uid[:len(name)] = np.frombuffer(name, np.uint8)
        fin = os.path.join(sub, "in.npz")
        np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=orc.theta, uid=uid, gamma=GAMMA, lam=LAM, seed=1234,
                 noise=noise, perms=perms, lr=LR, cr=CR, **{"ref_" + k: ro[k] for k in ("obs", "actions", "values", "neglogp", "returns")})
        env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="1", PPO_HIP_PEER_STATS=mode)
        res.append(run_workers(sub, world, fin, env))
    for a, b in zip(*res):
        for k in ("ro_obs", "ro_actions", "ro_values", "ro_neglogp", "ro_rewards", "ro_returns", "ro_dones", "obs_mean", "obs_var", "obs_count", "ret_mean", "ret_var", "ret_count"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    close(res[0][0]["obs_mean"], nz.obs_rms.mean, rtol=1e-5, atol=1e-6)
    assert float(res[0][0]["obs_count"]) == nz.obs_rms.count


@pytest.mark.gpu
@pytest.mark.parametrize("peer", [False, True])
@pytest.mark.parametrize("world,O,E,T", [(2, 64, 512, 3), (4, 128, 1024, 2)])
def test_wide_observation_statistics_under_data_parallelism(tmp_path, world, O, E, T, peer):
    """Observations a multiple of 64 wide with >= 256 environments per rank: norm_batch_kernel's column-group job (obs_cgroup_job) PUBLISHES each rank's batch moments
    from its group finishers instead of merging them (table all-reduce, or the peer slots written by the job's last group), norm_finalize_kernel combines the ranks.  The
    rollout shards and the running statistics against the single-process oracle over the union; statistics bit-identical on every rank."""
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    hidden, nmb, epochs = (256, 256), 4, 1
    orc = o.Oracle(O, 18, list(hidden)); orc.init_orthogonal(15)
    rng = np.random.RandomState(53)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    nz = o.Normalizer(E, O)
    ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
    El = E // world; Bl = El * T
    perms = np.stack([np.stack([rng.permutation(Bl).astype(np.int32) for _ in range(epochs)]) for _ in range(world)])
    uid = np.zeros(128, np.uint8)
    name = ("/ppo_dp_wide_%d_%d" % (os.getpid(), rng.randint(1 << 30))).encode()
    uid[:len(name)] = np.frombuffer(name, np.uint8)
    fin = os.path.join(tmp, "in.npz")
    np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=orc.theta, uid=uid, gamma=GAMMA, lam=LAM, seed=1234, O=O,
             noise=noise, perms=perms, lr=LR, cr=CR, **{"ref_" + k: ro[k] for k in ("obs", "actions", "values", "neglogp", "returns")})
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="1" if peer else "0")
    outs = run_workers(tmp, world, fin, env)
    for r, out in enumerate(outs):
        sl = slice(r * El, (r + 1) * El)
        for f in ("obs", "actions", "values", "neglogp", "rewards", "returns"):
            close(out["ro_" + f], ro[f][:, sl], rtol=2e-4, atol=2e-5, msg="rank %d %s" % (r, f))
        close(out["obs_mean"], nz.obs_rms.mean, rtol=1e-5, atol=1e-6); close(out["obs_var"], nz.obs_rms.var, rtol=1e-5)
        assert float(out["obs_count"]) == nz.obs_rms.count and float(out["ret_count"]) == nz.ret_rms.count
    for k in ("obs_mean", "obs_var", "ret_mean", "ret_var"):
        for out in outs[1:]:
            np.testing.assert_array_equal(outs[0][k], out[k])

