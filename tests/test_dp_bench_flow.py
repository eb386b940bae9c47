"""The multi-rank bench flow and the peer path's failure drill, on the ONE test GPU: two processes share device 0, RCCL (which
refuses two ranks per device) is swapped for tests/fake_rccl through PPO_RCCL_LIBRARY.  What is checked is everything a real
8-GPU run of bench.py depends on besides the links themselves: the launcher (torch.distributed.run, started before any GPU call of
its children), rendezvous on 127.0.0.1, the unique-id broadcast, both exchange paths, weak and strong scaling, the ONE JSON line of
rank 0, the replica digest -- and that a peer which stops answering surfaces as an error within PPO_HIP_PEER_TIMEOUT_MS, not a hang."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests.test_dp_two_ranks import build_fake_rccl
from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("collective,scaling,config", [("rccl", "weak", "cfg4"), ("peer", "weak", "cfg4"), ("auto", "weak", "cfg4"),
                                                       ("rccl", "strong", "cfg4"), ("peer", "strong", "cfg3"),
                                                       # BASELINE configs[4] (bf16): auto also times the gradient-bucket form of the exchange (ppo_dist_bucketed)
                                                       ("auto", "strong", "cfg5")])
def test_two_rank_bench_flow(tmp_path, collective, scaling, config):
    fake = build_fake_rccl(str(tmp_path))
    port = 29500 + (os.getpid() + hash((collective, scaling)) % 97) % 400
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", TMPDIR="/tmp")
    if config == "cfg5":
        env["PPO_HIP_PEER_REDUCE"] = "0"       # (two ranks of this shape on ONE device: the peer path's 19 MB pushes of both ranks wait on each other's workgroups -- a shared-device artefact)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", config, "--collective", collective, "--scaling", scaling]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["value"] > 0
    c = d["collectives"]
    assert c["replicas_bit_identical"] is True
    if collective in ("rccl", "peer"):
        assert c["used"] == collective
    else:
        assert set(c["auto_probe_ms_per_step"]) == ({"rccl", "rccl+buckets"} if config == "cfg5" else {"peer", "rccl"})
        assert c["used"] == min(c["auto_probe_ms_per_step"], key=c["auto_probe_ms_per_step"].get)
    assert "cpu_baseline" not in d                                # rank 0 at N = 1 only


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,extra", [(2, ["--config", "cfg4"]),
                                        # BASELINE configs[3] as written: 1024 envs over 8 GPUs = 128 envs x 64 steps, 256 minibatch rows per rank
                                        (8, ["--config", "cfg4", "--scaling", "strong"]),
                                        (4, ["--config", "cfg3", "--scaling", "strong", "--collective", "peer"])])
def test_bench_starts_its_own_ranks(tmp_path, gpus, extra):
    """`python bench.py --gpus N` with NO launcher in the command line (the form the driver's own command has): bench.py starts
    torch.distributed.run as a child before touching the GPU, relays rank 0's ONE JSON line and exits with the child's code.  The line
    says what the ranks were: ncclCommCount, every rank's device ordinal + PCI bus id, the collective library."""
    fake = build_fake_rccl(str(tmp_path))
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # stdout carries the JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "dp%d" % gpus
    c = d["collectives"]
    assert c["replicas_bit_identical"] is True and c["rccl_nranks"] == gpus
    assert [x["rank"] for x in c["devices"]] == list(range(gpus)) and all(x["pci_bus_id"] for x in c["devices"])
    assert len({x["pid"] for x in c["devices"]}) == gpus                          # one process per rank
    assert c["distinct_devices"] == 1                                             # (the one test GPU; a real node shows `gpus`)
    assert os.path.samefile(c["library"], fake)
    if "strong" in extra:
        E = {"cfg4": 1024, "cfg3": 4096}[extra[1]]
        assert d["scaling"] == "strong" and d["config"]["n_envs_per_gpu"] == E // gpus


def test_bench_refuses_more_ranks_than_devices_without_a_stand_in():
    """CPU-runnable: with no stand-in library and fewer devices than ranks, `bench.py --gpus N` says so in one JSON line and exits
    non-zero instead of handing RCCL two ranks per device."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PPO_RCCL_LIBRARY")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 2
    assert "RCCL needs one device per rank" in json.loads(out.stdout.strip().splitlines()[-1])["error"]


@pytest.mark.gpu
def test_a_peer_that_stops_answering_is_an_error_not_a_hang(tmp_path):
    """Rank 1 attaches, then never joins a collective again.  Rank 0's first exchange (the running-statistics table of its first env
    step) gives up after PPO_HIP_PEER_TIMEOUT_MS: the error surfaces from the rollout / update call and the process exits non-zero,
    well inside the test's time limit; nothing is re-executed, the device is usable afterwards."""
    tmp = str(tmp_path)
    fake = build_fake_rccl(tmp)
    E, T, nmb, epochs = 16, 8, 4, 1
    orc = o.Oracle(18, 18, [64, 64]); orc.init_orthogonal(1)
    rng = np.random.RandomState(5)
    uid = np.zeros(128, np.uint8)
    name = ("/ppo_dp_drill_%d_%d" % (os.getpid(), rng.randint(1 << 30))).encode()
    uid[:len(name)] = np.frombuffer(name, np.uint8)
    noise = rng.normal(size=(T, E, 18)).astype(np.float32)
    z = np.zeros((T, E), np.float32)
    fin = os.path.join(tmp, "in.npz")
    np.savez(fin, hidden=np.array([64, 64]), E=E, T=T, nmb=nmb, epochs=epochs, theta=orc.theta, uid=uid, gamma=0.99, lam=0.95, seed=1234, noise=noise,
             perms=np.stack([np.stack([rng.permutation(E // 2 * T).astype(np.int32) for _ in range(epochs)]) for _ in range(2)]), lr=3e-4, cr=0.2,
             ref_obs=np.zeros((T, E, 18), np.float32), ref_actions=np.zeros((T, E, 18), np.float32), ref_values=z, ref_neglogp=z, ref_returns=z)
    env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="1", PPO_TEST_DRILL="1", PPO_HIP_PEER_TIMEOUT_MS="400")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), "2", fin, os.path.join(tmp, "out%d.npz" % r)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=180)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the drill hung")
    assert procs[1].returncode == 0, logs[1][-2000:]
    assert procs[0].returncode != 0 and "gave up waiting" in logs[0], logs[0][-2000:]
    assert time.time() - t0 < 120
    # the device is fine afterwards
    import ppo_cpp_amd
    g = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); g.init_orthogonal(0)
    assert np.isfinite(g.value(np.zeros((4, 18), np.float32))).all()
    g.close()
