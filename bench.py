#!/usr/bin/env python3
"""bench.py -- PPO env-steps/s of the rollout-collect + minibatch-update hot path on N MI355X.

One "step" = one full PPO update of BASELINE.json configs[2] per rank: collect B = n_envs*n_steps transitions with the
18-obs/18-act MLP [256,256] policy (on-device seeded synthetic env of the env_mock shape, inputs resident in HBM),
then noptepochs x nminibatches clipped-surrogate train steps (fwd + loss + backward + global-norm clip + Adam).
`value` = total env-steps of all ranks / wall time (the reference's own `fps`, ppo2/ppo2.hpp:337-343).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...        (starts its N ranks itself, as a child torch.distributed.run; one rank per GPU over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this driver

CONFIGS = {
    # name: (n_envs, n_steps, hidden, obs, act, nminibatches, noptepochs)   -- SURVEY section 8 table
    "cfg3": dict(n_envs=4096, n_steps=16, hidden=[256, 256], obs=18, act=18, nminibatches=32, noptepochs=10,
                 desc="env_mock-shaped synthetic env, 4096 envs x 16 steps (B=65536), MLP [256,256], 32 minibatches x 10 epochs"),
    # the reference's OTHER real observation shape (observe_velocities: 36 observations, env/hexapod_closed_loop_env.hpp:20) on configs[2]'s workload
    "cfg3o36": dict(n_envs=4096, n_steps=16, hidden=[256, 256], obs=36, act=18, nminibatches=32, noptepochs=10,
                    desc="configs[2]'s workload with the reference's 36-observation hexapod shape: 4096 envs x 16 steps, MLP [256,256], 32 minibatches x 10 epochs"),
    "cfg2": dict(n_envs=1, n_steps=2048, hidden=[64, 64], obs=18, act=18, nminibatches=32, noptepochs=10,
                 desc="1 env x 2048 steps, MLP [64,64] (the reference's own command line; one resident launch per epoch)"),
    "cfg4": dict(n_envs=1024, n_steps=64, hidden=[64, 64], obs=18, act=18, nminibatches=32, noptepochs=10,
                 desc="1024 envs x 64 steps, MLP [64,64]"),
    # configs[1] / configs[3] with the hexapod's 36-observation shape (observe_velocities, env/hexapod_closed_loop_env.hpp:20,61-72)
    "cfg2o36": dict(n_envs=1, n_steps=2048, hidden=[64, 64], obs=36, act=18, nminibatches=32, noptepochs=10,
                    desc="1 env x 2048 steps, MLP [64,64], 36 observations"),
    "cfg4o36": dict(n_envs=1024, n_steps=64, hidden=[64, 64], obs=36, act=18, nminibatches=32, noptepochs=10,
                    desc="1024 envs x 64 steps, MLP [64,64], 36 observations"),
    "cfg5": dict(n_envs=8192, n_steps=16, hidden=[1024, 1024, 1024], obs=256, act=64, nminibatches=32, noptepochs=10,
                 dtype="bf16",
                 desc="synthetic 256-obs/64-act env, 8192 envs x 16 steps, MLP [1024,1024,1024], bf16 MFMA operands / fp32 accumulate, master weights and Adam"),
    "cfg5f32": dict(n_envs=8192, n_steps=16, hidden=[1024, 1024, 1024], obs=256, act=64, nminibatches=32, noptepochs=10,
                    desc="synthetic 256-obs/64-act env, 8192 envs x 16 steps, MLP [1024,1024,1024] in exact fp32 (two-tile LDS layout)"),
}
BASELINE_INDEX = {"cfg2": 1, "cfg2o36": 1, "cfg3": 2, "cfg3o36": 2, "cfg4": 3, "cfg4o36": 3, "cfg5": 4, "cfg5f32": 4}
LR, CR, GAMMA, LAM = 3.93141e-4, 0.161023, 0.99, 0.95      # README.md:70-81, ppo2.cpp:215-217
PEAK_F32_MFMA_TFLOPS = 157.3                                # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0                              # MI355X_MICROARCH.md: bf16 MFMA dense peak (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0


def flops_per_row(O, A, hidden):
    """(forward, dX backward, dW backward) FLOP per row, both towers (SURVEY section 8 table footnote)."""
    dims = [O] + list(hidden)
    tower = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    fwd = 2 * tower + hidden[-1] * A + hidden[-1]
    first = 2 * O * hidden[0]
    dx = fwd - first
    return 2 * fwd, 2 * dx, 2 * fwd


def granted_cores():
    """host cores this process may actually use: the affinity mask cut by the cgroup CPU quota (a container often SEES far more)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def vectorised_update_time(cfg, budget_s):
    """One WHOLE PPO update of `cfg` on the vectorised CPU port (oracle/numpy_port.py: NumPy expressions over BLAS sgemm; checked
    against the C oracle in tests/test_oracle.py), with whatever BLAS thread count this process was started with.  Times a bounded
    sample -- a few policy steps at the full n_envs rows, a few COMPLETE train steps (forward, loss, backward, clip, Adam over all
    parameters) at the full minibatch rows, one GAE scan, the running-statistics updates -- and extrapolates to T policy steps +
    epochs x minibatches train steps.  The synthetic env itself is not charged (the GPU side's env kernel is ~1 % of its collect)."""
    from oracle import oracle as o
    from oracle import numpy_port as npp
    E, T, nmb, ep = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"], cfg["noptepochs"]
    B = E * T; M = B // nmb
    O, A = cfg["obs"], cfg["act"]
    orc = o.Oracle(O, A, cfg["hidden"]); orc.init_orthogonal(0)
    P = npp.NumpyPPO(orc)
    rng = np.random.RandomState(0)
    f_fwd, f_dx, f_dw = flops_per_row(O, A, cfg["hidden"])
    Es = int(max(1, min(E, 4.0e9 // f_fwd))); Ms = int(max(2, min(M, 4.0e9 // (f_fwd + f_dx + f_dw))))     # (cfg5: 418 / 144 rows; cfg3: all)
    obs = rng.uniform(-1, 1, (Es, O)).astype(np.float32); noise = rng.normal(size=(Es, A)).astype(np.float32)

    def timed(fn, share, cap):
        fn()                                                 # warm: page in, BLAS thread pool up
        t0 = time.perf_counter(); n = 0
        while n < 1 or (time.perf_counter() - t0 < share * budget_s and n < cap):
            fn(); n += 1
        return (time.perf_counter() - t0) / n, n

    t_step, n_step = timed(lambda: P.step(obs, noise), 0.2, 64)
    mobs = rng.uniform(-1, 1, (Ms, O)).astype(np.float32)
    act, v, nlp = P.step(mobs, rng.normal(size=(Ms, A)).astype(np.float32))
    ret = (v + rng.normal(size=Ms)).astype(np.float32); adv = ((ret - v) - (ret - v).mean()) / ((ret - v).std() + 1e-8)
    t_train, n_tr = timed(lambda: P.train_step(LR, CR, mobs, act, adv.astype(np.float32), ret, nlp, v), 0.7, 256)
    Eg = min(E, 8192)
    rw = rng.normal(size=(T, Eg)).astype(np.float32); dn = (rng.rand(T, Eg) < 0.01).astype(np.float32)
    t_gae, _ = timed(lambda: npp.gae(rw, rw, dn, rw[0], dn[0], GAMMA, LAM), 0.05, 8)
    ob = rng.uniform(-1, 1, (Eg, O)).astype(np.float32)
    t_rs, _ = timed(lambda: npp.running_update(np.zeros(O, np.float32), np.ones(O, np.float32), 1.0, ob), 0.05, 16)
    t_update = T * (t_step * (E / Es) + t_rs * (E / Eg)) + t_gae * (E / Eg) + ep * nmb * t_train * (M / Ms)
    return {"value": B / t_update, "unit": "env-steps/s", "update_samples_per_s": ep * B / (ep * nmb * t_train * (M / Ms)),
            "t_policy_step_ms": 1e3 * t_step * (E / Es), "t_train_step_ms": 1e3 * t_train * (M / Ms),
            # what the timed statements cover of one update's algorithmic FLOPs (dense products of both towers,
            # forward + dX + dW); the O(rows x A) loss arithmetic, clip and Adam are timed too but are not in that FLOP count
            "covers": 1.0,
            "covers_what": "whole update: T policy steps (forward, sample, neglogp), running statistics, GAE, and epochs x minibatches COMPLETE train steps "
                           "(forward, loss, backward incl. TanhGrad and all weight / bias gradients, global-norm clip, Adam over all %d parameters); the env is not charged" % orc.P,
            "sample": "%d policy steps at %d rows + %d train steps at %d rows, extrapolated to %d steps + %d train steps per update"
                      % (n_step, Es, n_tr, Ms, T, ep * nmb) + ("" if (Es, Ms) == (E, M) else " (rows scaled to %d / %d)" % (E, M))}


def _rows_worker(cfg, n_proc, budget_s, t_start):
    """one of n_proc row-parallel workers (1 BLAS thread each): policy steps on n_envs / n_proc rows, forward + loss + backward on
    minibatch rows / n_proc; all workers start together so that the cores are loaded at once"""
    from oracle import oracle as o
    from oracle import numpy_port as npp
    E, T, nmb = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"]
    M = E * T // nmb
    O, A = cfg["obs"], cfg["act"]
    orc = o.Oracle(O, A, cfg["hidden"]); orc.init_orthogonal(0)
    P = npp.NumpyPPO(orc)
    rng = np.random.RandomState(1)
    f_fwd, f_dx, f_dw = flops_per_row(O, A, cfg["hidden"])
    # a worker's share of a call: ceil(rows / n_proc) -- never less than ONE environment / two minibatch rows (with fewer rows than processes the
    # slowest process still does a whole row: no credit for splitting what cannot be split)
    e_share = max(1, -(-E // n_proc)); m_share = max(2, -(-M // n_proc))
    e_rows = int(max(1, min(e_share, 4.0e9 // f_fwd))); m_rows = int(max(2, min(m_share, 4.0e9 // (f_fwd + f_dx + f_dw))))
    obs = rng.uniform(-1, 1, (e_rows, O)).astype(np.float32); noise = rng.normal(size=(e_rows, A)).astype(np.float32)
    mobs = rng.uniform(-1, 1, (m_rows, O)).astype(np.float32)
    act, v, nlp = P.step(mobs, rng.normal(size=(m_rows, A)).astype(np.float32))
    ret = (v + rng.normal(size=m_rows)).astype(np.float32); adv = (((ret - v) - (ret - v).mean()) / ((ret - v).std() + 1e-8)).astype(np.float32)
    P.step(obs, noise); P.loss_grad(mobs, act, adv, ret, nlp, v, CR)                    # warm
    while time.time() < t_start:
        time.sleep(0.002)

    def timed(fn, share, cap):
        t0 = time.perf_counter(); n = 0
        while n < 1 or (time.perf_counter() - t0 < share * budget_s and n < cap):
            fn(); n += 1
        return (time.perf_counter() - t0) / n

    t_step = timed(lambda: P.step(obs, noise), 0.25, 64) * (e_share / e_rows)
    t_lg = timed(lambda: P.loss_grad(mobs, act, adv, ret, nlp, v, CR), 0.75, 256) * (m_share / m_rows)
    return {"t_step": t_step, "t_loss_grad": t_lg, "e_rows": e_rows, "m_rows": m_rows}


def cpu_all_cores_leg(cfg, name, n_proc, budget_s):
    """The vectorised port ROW-PARALLEL over all granted cores: n_proc processes of one BLAS thread each take 1 / n_proc of every call's rows
    (a policy step's environments, a minibatch's rows), started together; a train step = the slowest worker's forward + loss + backward on its
    rows + the cross-worker sum of the n_proc gradient vectors + clip + Adam (both timed here, in one process, and charged in full)."""
    import subprocess
    from oracle import oracle as o
    from oracle import numpy_port as npp
    E, T, nmb, ep = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"], cfg["noptepochs"]
    B = E * T
    t_start = time.time() + 6.0
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", name, "rows", str(n_proc), repr(budget_s), repr(t_start)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT) for _ in range(n_proc)]
    res = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=120 + 4 * budget_s)
            res.append(json.loads(out.strip().splitlines()[-1]))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    orc = o.Oracle(cfg["obs"], cfg["act"], cfg["hidden"]); orc.init_orthogonal(0)
    P = npp.NumpyPPO(orc)
    grads = np.random.RandomState(0).normal(size=(n_proc, orc.P)).astype(np.float32)
    t0 = time.perf_counter()
    for _ in range(20):
        g = grads.sum(0, dtype=np.float32); P.clip_adam(g, LR)
    t_red = (time.perf_counter() - t0) / 20
    t_step = max(r["t_step"] for r in res); t_lg = max(r["t_loss_grad"] for r in res)
    t_update = T * t_step + ep * nmb * (t_lg + t_red)
    # how evenly the box served the workers (a process that shares its core with somebody else's job is several times slower: `value` counts the
    # slowest, as a synchronous row-parallel step must; the median tells what the cores could do)
    import statistics
    per_worker = sorted(B / (T * r["t_step"] + ep * nmb * (r["t_loss_grad"] + t_red)) for r in res)
    return {"value": B / t_update, "unit": "env-steps/s", "cores": n_proc, "covers": 1.0,
            "workers_env_steps_per_s": {"min": per_worker[0], "median": statistics.median(per_worker), "max": per_worker[-1]},
            "note": "box-dependent (cores granted vs cores busy elsewhere cannot be known): the single-thread leg is the stable figure",
            "t_policy_step_ms": 1e3 * t_step, "t_train_step_ms": 1e3 * (t_lg + t_red), "t_reduce_clip_adam_ms": 1e3 * t_red,
            "sample": "%d processes x 1 BLAS thread, each 1/%d of the rows (%d policy-step rows, %d minibatch rows), slowest process; gradient sum + clip + Adam timed in one process"
                      % (n_proc, n_proc, res[0]["e_rows"], res[0]["m_rows"])}


def cpu_vectorised_leg(name, threads, budget_s):
    """the vectorised port in a CHILD process whose BLAS thread count is fixed before NumPy loads (OPENBLAS / OMP / MKL _NUM_THREADS)"""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", name, str(threads), repr(budget_s)], capture_output=True, text=True,
                         timeout=60 + 6 * budget_s, env=env, cwd=ROOT)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    r["cores"] = threads
    return r


def cpu_baseline(cfg, name="cfg3", budget_s=20.0):
    """CPU legs of the same workload on the GPU box's host cores (kind "port": the reference needs TensorFlow-C++ and cannot be built here):
      value / cores = 1 : the vectorised port (NumPy over BLAS sgemm), ONE thread, a WHOLE update (`covers` 1.0);
      all_cores          : the same port row-parallel over the cores this process is granted (affinity mask cut by the cgroup quota): one
                           process of one BLAS thread per core, the gradient sum + clip + Adam charged on top;
      scalar_port        : the oracle's C restatement (scalar loops, double accumulators), one thread -- the checker itself, on record."""
    from oracle import oracle as o
    E, T, nmb, ep = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"], cfg["noptepochs"]
    B = E * T; M = B // nmb
    one = cpu_vectorised_leg(name, 1, 0.5 * budget_s)
    n = granted_cores()
    try:
        allc = cpu_all_cores_leg(cfg, name, n, 0.25 * budget_s) if n > 1 else dict(one)
    except Exception as e:                                   # the single-thread leg is the contract; this one is extra
        allc = {"error": repr(e)}
    # the scalar port on a bounded sample: about 1 GFLOP per call (it runs ~2-3 GFLOP/s); per-row cost is size independent
    f_fwd, f_dx, f_dw = flops_per_row(cfg["obs"], cfg["act"], cfg["hidden"])
    Es = int(max(16, min(E, 1.0e9 // f_fwd))); Ms = int(max(16, min(M, 1.0e9 // (f_fwd + f_dx + f_dw))))
    orc = o.Oracle(cfg["obs"], cfg["act"], cfg["hidden"])
    orc.init_orthogonal(0)
    rng = np.random.RandomState(0)
    obs = rng.uniform(-1, 1, (Es, cfg["obs"])).astype(np.float32)
    noise = rng.normal(size=(Es, cfg["act"])).astype(np.float32)
    t0 = time.perf_counter(); n_step = 0
    while n_step < 1 or (time.perf_counter() - t0 < 0.03 * budget_s and n_step < 4):
        a, v, nlp = orc.step(obs, noise); n_step += 1
    t_step = (time.perf_counter() - t0) / n_step * (E / Es)
    mobs = rng.uniform(-1, 1, (Ms, cfg["obs"])).astype(np.float32)
    act, v, nlp = orc.step(mobs, rng.normal(size=(Ms, cfg["act"])).astype(np.float32))
    ret = (v + rng.normal(size=Ms)).astype(np.float32)
    adv = o.adv_normalize(ret, v)
    t0 = time.perf_counter(); n_tr = 0
    while n_tr < 1 or (time.perf_counter() - t0 < 0.17 * budget_s and n_tr < 16):
        orc.train_step(LR, CR, mobs, act, adv, ret, nlp, v); n_tr += 1
    t_train = (time.perf_counter() - t0) / n_tr * (M / Ms)
    scalar = {"value": B / (T * t_step + ep * nmb * t_train), "unit": "env-steps/s", "cores": 1,
              "kind": "port (oracle's C restatement: scalar loops, double accumulators)",
              "sample": "%d policy steps at %d rows + %d train steps at %d rows, rows scaled to %d / %d" % (n_step, Es, n_tr, Ms, E, M)}
    out = dict(one) if "value" in one else dict(scalar)
    out["kind"] = "port"
    out["kind_detail"] = "vectorised port (oracle/numpy_port.py: NumPy expressions over BLAS sgemm), 1 thread" if "value" in one else scalar["kind"]
    out["cores"] = 1
    out["all_cores"] = allc
    out["scalar_port"] = scalar
    return out


def count_gpus_from_sysfs():
    """GPU nodes of /sys/class/kfd/kfd/topology (a node with simd_count > 0 is a GPU; CPUs have 0), cut by ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES when they are plain index lists.  No topology directory = no amdgpu compute driver on this machine = 0 devices; None only when the
    files exist and cannot be read"""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return 0
    try:
        for f in files:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and all(x.strip().isdigit() for x in v.split(",") if x.strip()):
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` -- as a CHILD process (never exec: this
    process stays a plain parent that has not touched the GPU), relay rank 0's single JSON line to stdout, everything else to stderr,
    and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if not os.environ.get("PPO_RCCL_LIBRARY"):
        # RCCL takes ONE rank per device.  (PPO_RCCL_LIBRARY = a stand-in such as tests/fake_rccl lets N ranks share a device: a dry
        # run of the flow, not a measurement.)  The devices are counted from the kernel driver's topology files -- no HIP / torch call, so this
        # parent really never touches the GPU.
        ndev = count_gpus_from_sysfs()
        if ndev is not None and ndev < n:
            print(json.dumps({"error": "bench.py --gpus %d: this node exposes %d HIP device(s); RCCL needs one device per rank" % (n, ndev)}), flush=True)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")                   # (what the launcher would set itself, without its warning)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    try:
        for line in child.stdout:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            (sys.stdout if line.startswith("{") else sys.stderr).flush()
        return child.wait()
    except BaseException:
        child.kill()
        child.wait()
        raise


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":          # child of cpu_vectorised_leg: CPU only, BLAS threads fixed by its environment
        if sys.argv[3] == "rows":
            print(json.dumps(_rows_worker(CONFIGS[sys.argv[2]], int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6]))))
        else:
            print(json.dumps(vectorised_update_time(CONFIGS[sys.argv[2]], float(sys.argv[4]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] side measurement")
    ap.add_argument("--host-env", action="store_true", help="also time the Env-on-host path (PCIe inclusive)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank runs the config's n_envs; strong: the config's n_envs are divided over the ranks "
                         "(BASELINE configs[3] as written: 1024 envs in total over 8 GPUs)")
    ap.add_argument("--collective", default="rccl", choices=["auto", "peer", "rccl"],
                    help="data-parallel exchange: ncclAllReduce (default: the only form that has run across physical devices), the "
                         "one-shot peer all-reduce over IPC-mapped buffers, or auto = when the peer probe passes on every rank, TIME both over "
                         "a few steps and keep the faster one for the timed region (both timings go into `collectives`)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                     # BEFORE the package is imported or any GPU call is made
    from ppo_cpp_amd import dist as ppodist
    rank, world, local_rank = ppodist.env_rank_world()
    if world != args.gpus:
        args.gpus = world                                    # under a launcher the launcher's world size is the truth

    import ppo_cpp_amd
    dist = None
    device = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only; data path = RCCL in libppo_hip
        device = -1                                                       # the library picks LOCAL_RANK % (its own device count)

    E, T, nmb, ep = cfg["n_envs"], cfg["n_steps"], cfg["nminibatches"], cfg["noptepochs"]
    if args.scaling == "strong":
        if E % world:
            sys.exit("--scaling strong: %d envs do not divide over %d ranks" % (E, world))
        E //= world                                                        # per-rank share; minibatch rows per rank shrink with it
    B = E * T; M = B // nmb
    bf16 = cfg.get("dtype") == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    g = ppo_cpp_amd.PPOHip(cfg["obs"], cfg["act"], cfg["hidden"], device=device, compute_dtype=1 if bf16 else 0)
    g.init_orthogonal(0)                                                   # same seed on every rank: replicated weights
    if world > 1:
        g.dist_init(world, rank, ppodist.broadcast_unique_id(dist, rank, ppo_cpp_amd.PPOHip.dist_unique_id))
        if args.collective != "rccl" and world <= 8:
            peer_ok = g.dist_peer_attach(ppodist.allgather_bytes(dist, g.dist_peer_export(), 64))
            if args.collective == "peer" and not peer_ok:
                sys.exit("--collective peer: the peer all-reduce probe failed (or its region is not fine-grained memory)")
    g.norm_init(E, GAMMA)
    g.rollout_alloc(E, T)
    env0 = ppodist.env_offset(E, rank)                                     # every rank owns its own E environments (global ids rank*E ..)

    def one_step(i, first=False):
        g.collect_synthetic(1234, GAMMA, LAM, None, env0=env0, step0=i * T, first=first)
        return g.update(LR, CR, ep, nmb, None, seed=1000 + i, want_rows=False)[1]

    def barrier():
        g.sync()
        if dist is not None:
            dist.barrier()

    probe = None
    peer_attached = dist is not None and g.dist_peer_active()
    if dist is not None and args.collective == "auto" and (peer_attached or bf16):
        # every exchange form over a few steps each (first one warm / captured), the fastest one runs the timed region: the peer regions (when they attached),
        # ncclAllReduce, and -- bf16 path -- ncclAllReduce of the gradient in layer BUCKETS on a second stream (the library's default there, ppo_dist_bucketed)
        probe = {}
        one_step(0, first=True)
        modes = (("peer",) if peer_attached else ()) + (("rccl", "rccl+buckets") if bf16 else ("rccl",))

        def select(mode):
            if peer_attached:
                g.dist_peer_enable(mode == "peer")
            if bf16:
                g.dist_bucketed(mode == "rccl+buckets")
        for mode in modes:
            select(mode)
            one_step(1)
            barrier()
            t1 = time.perf_counter()
            for i in range(2):
                one_step(2 + i)
            g.sync()
            probe[mode] = 1e3 * ppodist.allreduce_max(dist, time.perf_counter() - t1) / 2
            barrier()
        select(min(probe, key=probe.get))
    for i in range(max(args.warmup, 1)):
        losses = one_step(i, first=(i == 0 and probe is None))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = one_step(args.warmup + i)
    g.sync()
    dt = time.perf_counter() - t0
    barrier()
    if dist is not None:
        dt = ppodist.allreduce_max(dist, dt)
    collectives = None
    if dist is not None:
        # the replicas must have stayed bit-identical (same reduced gradient in the same order on every rank): checked, and fatal
        import hashlib
        used = "peer" if g.dist_peer_active() else ("rccl+buckets" if (bf16 and (probe is None or min(probe, key=probe.get) == "rccl+buckets")) else "rccl")
        digests = ppodist.allgather_bytes(dist, hashlib.sha256(g.get_flat(0).tobytes()).digest(), 32)
        info = g.dist_info()
        mine = json.dumps({"rank": rank, "local_rank": local_rank, "device": info["device"], "pci_bus_id": info["pci_bus_id"], "pid": os.getpid()}).encode()
        devices = [json.loads(b.rstrip(b" ").decode()) for b in ppodist.allgather_bytes(dist, mine.ljust(160), 160)]
        collectives = {"used": used, "graph_captured": g.dist_graph_collectives(), "replicas_bit_identical": len(set(digests)) == 1,
                       used: {"ms_per_step": 1e3 * dt / args.steps},
                       # what lets a reader check the ranks: the communicator's own rank count (ncclCommCount), every rank's HIP ordinal +
                       # PCI bus id (distinct devices on a real node; all the same on a one-GPU dry run), the library behind the nccl* calls
                       "rccl_nranks": info["comm_nranks"], "devices": devices, "library": info["library"],
                       "distinct_devices": len({d["pci_bus_id"] for d in devices})}
        if probe is not None:
            collectives["auto_probe_ms_per_step"] = probe
        if not collectives["replicas_bit_identical"]:
            if rank == 0:
                print(json.dumps({"error": "replicas diverged: the weights are not bit-identical across the ranks", "collectives": collectives}), flush=True)
            g.close()
            sys.exit(3)

    # per-kernel device time of the same workload, HIP events on the handle's stream, right after the timed region
    g.prof_enable(True)
    t_c0 = time.perf_counter()
    g.collect_synthetic(1234, GAMMA, LAM, None, env0=env0, step0=(args.warmup + args.steps) * T, first=False)
    t_collect = time.perf_counter() - t_c0
    g.update(LR, CR, ep, nmb, None, seed=999, want_rows=False)
    prof = g.prof_read()
    g.prof_enable(False)
    # the reference's own shape with minibatches of <= 64 rows: ONE launch per epoch runs all its train steps (narrow_epoch_kernel: forward, backward, weight
    # gradients, assembly, clip + Adam of every minibatch) -- the "train_fwd_bwd" class is then that launch
    epoch_fused = g.kernel_counts().get("narrow_epoch_kernel", 0) > 0

    # phase split of one un-profiled step
    t_a = time.perf_counter()
    g.collect_synthetic(1234, GAMMA, LAM, None, env0=env0, step0=(args.warmup + args.steps + 1) * T, first=False)
    t_b = time.perf_counter()
    g.update(LR, CR, ep, nmb, None, seed=998, want_rows=False)
    t_c = time.perf_counter()

    if world > 1:
        # orderly shutdown: every rank is past its last collective before any communicator is torn down
        barrier()
        g.close()
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return
    if rank != 0:
        return
    f_fwd, f_dx, f_dw = flops_per_row(cfg["obs"], cfg["act"], cfg["hidden"])
    kflops = {"train_fwd_bwd": (f_fwd + f_dx) * M, "weight_grad": f_dw * M, "policy_step": f_fwd * E}
    if epoch_fused:
        kflops["train_fwd_bwd"] = (f_fwd + f_dx + f_dw) * M * nmb
    kern = {k: {"avg_us": 1e3 * ms / n, "launches": n} for k, (ms, n) in prof.items() if n}
    dom = max((k for k in kern if k in kflops), key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    ach = kflops[dom] / (kern[dom]["avg_us"] * 1e-6) / 1e12
    # PMC passes and the profiler's kernel trace cannot run inside the timed process: both come from the committed files that
    # profiles/current.json names (written with the round's profiles), and the JSON line says so
    traffic = traffic_source = rocprof_avg_us = rocprof_source = None
    # kernel names of the dominant class, most specific first (the class "train_fwd_bwd" is train8_kernel on the 18-obs / [256,256] shape,
    # train_fwd_bwd_kernel on other wide fp32 shapes, narrow_train_kernel for nets <= 64 wide, the tanh GEMM on the bf16 path)
    # The timed class may be ONE kernel (fp32 paths) or a SEQUENCE of launches per train step (bf16 path: L tanh GEMMs, the head kernel, the loss
    # kernel, L TanhGrad GEMMs).  `members` = (kernel-name pattern, launches of it per launch of the class); the profile numbers of a
    # class are the sums over its members x their launches per step, so that traffic / rocprof_avg_us describe the same thing as
    # flop_per_launch and avg_us.  (For a single-kernel class that is just that kernel's row.)
    L = len(cfg["hidden"])
    if bf16:
        # (round 5: the hidden layers of a pass are ONE chained launch at this minibatch size; a profile taken with PPO_HIP_NO_BF16_CHAIN=1 holds the
        # launch-per-layer names instead -- `alt_members` below)
        alt_members = [("gemm_nt_bf16_kernel<4, 0>", L), ("bf16_heads_kernel", 1), ("bf16_loss_kernel", 1), ("gemm_nt_bf16_kernel<4, 1>", L)]
        members = {"train_fwd_bwd": [("gemm_chain_bf16_kernel<0>", 1), ("bf16_heads_kernel", 1), ("bf16_loss_kernel", 1), ("gemm_chain_bf16_kernel<1>", 1)],
                   "weight_grad": [("gemm_dw_bf16_kernel", 1)],
                   "policy_step": [("bf16_stage_kernel", 1), ("gemm_nt_bf16_kernel<4, 0>", L), ("bf16_heads_kernel", 1), ("bf16_sample_kernel", 1)]}[dom]
    else:
        first = {"train_fwd_bwd": ["train8_kernel", "train_fwd_bwd_kernel", "narrow_epoch_kernel", "narrow_train_kernel"],
                 "weight_grad": ["weight_grad_assemble_kernel", "weight_grad_kernel"],
                 "policy_step": ["policy_step_kernel", "narrow_rollout1_kernel", "narrow_rollout_kernel", "narrow_step_kernel"]}[dom]
        members = None                                       # resolved below: the first of these names the profile holds
        alt_members = None
    try:
        idx = json.load(open(os.path.join(ROOT, "profiles", "current.json"))).get(args.config, {})
        note = (" [profiled with %s]" % idx["env"]) if idx.get("env") else ""      # e.g. eager launches where the profiler cannot follow the update's graph
        if idx.get("hbm_traffic"):
            kk = json.load(open(os.path.join(ROOT, "profiles", idx["hbm_traffic"])))["kernels"]
            if members and dom == "train_fwd_bwd" and alt_members and not any(members[0][0] in k for k in kk):
                members = alt_members
            mem = members or [(n, 1) for n in first if any(n in k for k in kk)][:1]
            tot, ok = 0.0, bool(mem)
            for pat, per in mem:
                hit = [v for k, v in kk.items() if pat in k][:1]
                if not hit or "FETCH_SIZE" not in hit[0] or "WRITE_SIZE" not in hit[0]:
                    ok = False
                    break
                tot += per * (2.0 * hit[0]["FETCH_SIZE"] + hit[0]["WRITE_SIZE"]) * 1024.0
            if ok:
                traffic = tot
                traffic_source = "profiles/%s (offline rocprofv3 --pmc passes of this command; FETCH_SIZE doubled per MI355X_MICROARCH.md; %s)%s" % (
                    idx["hbm_traffic"], " + ".join("%d x %s" % (per, pat) for pat, per in mem), note)
        if idx.get("kernel_stats"):
            import csv
            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", idx["kernel_stats"]))))
            if members and dom == "train_fwd_bwd" and alt_members and not any(members[0][0] in r["Name"] for r in rows):
                members = alt_members
            mem = members or [(n, 1) for n in first if any(n in r["Name"] for r in rows)][:1]
            tot, ok = 0.0, bool(mem)
            for pat, per in mem:
                hit = [r for r in rows if pat in r["Name"]][:1]
                if not hit:
                    ok = False
                    break
                tot += per * float(hit[0]["AverageNs"]) / 1e3
            if ok:
                rocprof_avg_us = tot
                rocprof_source = "profiles/%s (rocprofv3 --kernel-trace --stats of this command; %s)%s" % (
                    idx["kernel_stats"], " + ".join("%d x %s" % (per, pat) for pat, per in mem), note)
    except Exception as e:
        traffic_source = "unavailable: %r" % (e,)
    step_flops = (f_fwd + f_dx + f_dw) * M
    step_us = sum(kern[k]["avg_us"] for k in ("train_fwd_bwd", "weight_grad", "grad_reduce", "adam") if k in kern)
    if epoch_fused:
        step_us = kern["train_fwd_bwd"]["avg_us"] / nmb
    out = {
        "metric": "PPO env-steps/s", "value": world * B * args.steps / dt, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d] (%s): %s" % (BASELINE_INDEX[args.config], args.config, cfg["desc"]), "n_envs_per_gpu": E, "n_steps": T,
                   "n_batch_per_gpu": B, "minibatch_rows_per_gpu": M, "parallelism": "dp%d" % world,
                   "env": "on-device seeded synthetic env (env_mock shape), rollout buffers resident in HBM"},
        "update_samples_per_s": world * ep * B / (t_c - t_b),
        "phase_ms": {"collect": 1e3 * (t_b - t_a), "update": 1e3 * (t_c - t_b)},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                     "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                     "rocprof_avg_us": rocprof_avg_us, "rocprof_source": rocprof_source,
                     "flop_per_launch": kflops[dom], "avg_us": kern[dom]["avg_us"],
                     "timing": "HIP events on the handle's stream around every launch of an eager pass right after the timed region: "
                               "an UPPER bound on kernel time (launch gaps included); the rocprofv3 kernel-trace of the same command is in profiles/",
                     # whole train step against the same peak, from the graph-replayed update phase (no launch gaps, device paced)
                     "step_frac": step_flops / ((t_c - t_b) / (ep * nmb)) / 1e12 / peak,
                     "train_step": {"flop": step_flops, "us_from_update_phase": 1e6 * (t_c - t_b) / (ep * nmb), "event_us_sum_upper_bound": step_us,
                                    "achieved": step_flops / ((t_c - t_b) / (ep * nmb)) / 1e12,
                                    **({"launches": "one narrow_epoch_kernel launch per epoch = %d train steps (the dominant class is that launch)" % nmb} if epoch_fused else {})}},
        "kernels": kern,
        "losses": [float(x) for x in losses],
    }
    if collectives is not None:
        out["collectives"] = collectives
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg, args.config)
    # (skipped under rocprofv3: instantiating a second handle's hipGraph in one traced process crashes the profiler)
    profiled = any("ROCPROF" in k for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and args.config == "cfg3" and not args.no_extra and not profiled:
        # BASELINE configs[1] (one environment, MLP [64,64], 2048 steps per update: launch-latency bound) measured in the
        # same run, so that both single-GPU configurations of the baseline are on record; `value` stays configs[2]
        g.close()
        c2 = CONFIGS["cfg2"]
        out["also"] = {}
        def cfg2_leg(fast):
            """configs[1] on the device env; fast = the opt-in 1-ulp Adam quotient (PPO_HIP_ADAM_FAST=1, read at ppo_create), else the default (correctly rounded: no deviation)"""
            if fast:
                os.environ["PPO_HIP_ADAM_FAST"] = "1"
            g2, res = None, None
            try:
                for attempt in range(2):                         # extra leg: never fatal for the contract line.  (A kernel whose workgroups meet inside a launch reports a
                    try:                                         # failed placement / residency check ONCE and the handle falls back to the launch per step: second attempt.)
                        if g2 is None:
                            g2 = ppo_cpp_amd.PPOHip(c2["obs"], c2["act"], c2["hidden"], device=device)
                            g2.init_orthogonal(0); g2.norm_init(c2["n_envs"], GAMMA); g2.rollout_alloc(c2["n_envs"], c2["n_steps"])
                        def step2(i, first=False):
                            g2.collect_synthetic(1234, GAMMA, LAM, None, env0=0, step0=i * c2["n_steps"], first=first)
                            g2.update(LR, CR, c2["noptepochs"], c2["nminibatches"], None, seed=2000 + i, want_rows=False)
                        step2(0, True); g2.sync()
                        t0 = time.perf_counter()
                        for i in range(3):
                            step2(1 + i)
                        g2.sync()
                        dt2 = (time.perf_counter() - t0) / 3
                        res = {"value": c2["n_envs"] * c2["n_steps"] / dt2, "unit": "env-steps/s", "ms_per_step": 1e3 * dt2, "steps": 3, "warmup": 1}
                        break
                    except Exception as e:
                        res = {"error": repr(e)}
            finally:
                if fast:
                    os.environ.pop("PPO_HIP_ADAM_FAST", None)
                if g2 is not None:
                    try:
                        g2.close()
                    except Exception:
                        pass
            return res
        leg = cfg2_leg(False)
        leg.update({"workload": c2["desc"], "adam": "correctly rounded quotient (the default since round 6: no deviation from the reference's arithmetic)"})
        fast = cfg2_leg(True)
        leg["with PPO_HIP_ADAM_FAST=1 (opt-in 1-ulp quotient)"] = {k: fast[k] for k in fast if k in ("value", "ms_per_step", "error")}
        out["also"]["BASELINE configs[1] (cfg2)"] = leg
        g = None
        try:
            # ... and the same configuration as the reference actually runs it: the ONE environment stepped on the HOST (Env::step
            # behind VecEnv / EnvNormalize, PPO2::learn), every env step a PCIe round trip
            from ppo_cpp_amd import hostapi
            r2 = hostapi.learn(c2["n_envs"], c2["n_steps"], c2["hidden"], n_updates=5, nminibatches=c2["nminibatches"], noptepochs=c2["noptepochs"],
                               lr=LR, cliprange=CR, gamma=GAMMA, lam=LAM)
            out["also"]["BASELINE configs[1] (cfg2), Env on the host"] = {
                "value": r2["env_steps_per_s"], "unit": "env-steps/s", "collect_ms": r2["collect_ms"], "update_ms": r2["update_ms"],
                "us_per_env_step": 1e3 * r2["collect_ms"] / c2["n_steps"],
                "note": "SeededEnvMock on the host; resident rollout kernel, actions / transitions through pinned memory"}
        except Exception as e:
            out["also"]["BASELINE configs[1] (cfg2), Env on the host"] = {"error": repr(e)}
    # PCIe-inclusive leg: on by default in the plain single-GPU run (same guard as above), or forced with --host-env
    if world == 1 and (args.host_env or (args.config == "cfg3" and not args.no_extra and not profiled)):
        # PCIe-inclusive: the reference's own stack (N x mock Env -> VecEnv -> EnvNormalize -> PPO2::learn) on the host,
        # actions D2H / observations H2D every env step.  Reported beside `value`, never as `value`.
        try:
            from ppo_cpp_amd import hostapi
            if g is not None:
                g.close()
            r = hostapi.learn(E, T, cfg["hidden"], n_updates=6, nminibatches=nmb, noptepochs=ep, lr=LR, cliprange=CR, gamma=GAMMA, lam=LAM)
            out["host_env"] = {"env_steps_per_s": r["env_steps_per_s"], "collect_ms": r["collect_ms"], "update_ms": r["update_ms"],
                               "collect_phase_ms": r["phase_ms"], "vec_env_pool": r["vec_env_pool"],
                               "note": "SeededEnvMock x %d behind the pooled VecEnv on the host cores, PCIe round trip per env step" % E}
        except Exception as e:                               # extra leg: never fatal for the contract line
            out["host_env"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
