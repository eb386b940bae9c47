"""The first run on more than one DEVICE, as ONE command (VERDICT r5 task 5b; reference ppo2.cpp:188-217 is the job being scaled).
Everything in DESIGN.md section 6 has only ever run as processes on ONE device; the day a node exists this script produces the whole curve and the checks that matter there:

  1. how many devices there are (a child process: the parent never touches the GPU, so every step below starts from a fresh process);
  2. RCCL at world N: `bench.py --gpus N --config cfg4 --collective rccl` for 2 steps -- ncclCommCount, distinct PCI bus ids, whether the capture probe let the
     collectives into the update's graph (`collectives.graph_captured`), replicas bit-identical;
  3. the peer attach probe at world N: the same with `--collective peer` (cross-DEVICE ordering of the write-through tile push: the first thing to look at when this fails);
  4. the curve: bench.py --gpus 1 / 2 / 4 / 8 for cfg3, cfg4 (strong: BASELINE configs[3] as written) and cfg5, `--collective rccl` then `--collective auto`;
  5. the C++ host layer: `ppo_cpp_hip --ranks N` (no Python in the job), replicas' checkpoints byte-identical;
  and writes ONE JSON (default gpurun_out/first_8gpu.json): per step the command, exit code, seconds, the parsed line; at the top `rccl_nranks`, `distinct_devices`,
  `replicas_bit_identical`, `values` {config: {N: env-steps/s}}.

usage: python tools/first_8gpu.py [--out FILE] [--max-gpus N] [--quick] [--stand-in LIBFAKE_RCCL.so]
  --stand-in: a one-GPU DRY RUN of the same flow (every rank on device 0, tests/fake_rccl behind the nccl* calls): what tests/test_tools.py runs on the test box.
  --quick: 2 steps per bench line and cfg4 only (the dry run's setting)."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cmd, env=None, timeout=1800):
    t0 = time.time()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        rc, out, err = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -9, (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout after %d s" % timeout
    return {"cmd": " ".join(cmd), "rc": rc, "seconds": round(time.time() - t0, 1), "stdout_tail": out[-1500:], "stderr_tail": err[-1500:]}, out


def bench_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "first_8gpu.json"))
    ap.add_argument("--max-gpus", type=int, default=8)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--stand-in", default=None)
    a = ap.parse_args()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if a.stand_in:
        env["PPO_RCCL_LIBRARY"] = os.path.abspath(a.stand_in)
    report = {"steps": [], "stand_in": bool(a.stand_in)}
    # 1. devices (torch.cuda.device_count() does not initialise the GPU on this image; it runs in a child anyway)
    st, out = child([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], env)
    report["steps"].append(dict(st, what="device count"))
    devices = int(out.strip().splitlines()[-1]) if st["rc"] == 0 and out.strip() else 0
    report["devices"] = devices
    top = a.max_gpus if a.stand_in else min(a.max_gpus, devices)
    worlds = [n for n in (1, 2, 4, 8) if n <= top]
    N = worlds[-1] if worlds else 0
    steps, warm = ("2", "1") if a.quick else ("20", "5")

    def bench(n, config, collective, extra=()):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", steps, "--warmup", warm, "--config", config, "--no-cpu-baseline", "--no-extra"]
        if n > 1:
            cmd += ["--collective", collective]
        st, out = child(cmd + list(extra), env)
        d = bench_line(out)
        st["line"] = d
        return st, d
    ok_all = True
    if N > 1:
        # 2. + 3. the two exchange paths at the largest world
        for collective, what in (("rccl", "RCCL at world %d (capture probe, ncclCommCount, distinct devices)" % N), ("peer", "peer attach probe at world %d" % N)):
            st, d = bench(N, "cfg4", collective, ["--scaling", "strong"])
            st["what"] = what
            report["steps"].append(st)
            c = (d or {}).get("collectives", {})
            if collective == "rccl":
                report["rccl_nranks"] = c.get("rccl_nranks")
                report["distinct_devices"] = c.get("distinct_devices")
                report["graph_collectives_rccl"] = c.get("graph_captured")
                report["collective_library"] = c.get("library")
            else:
                report["peer_path_used"] = c.get("used") == "peer"
            ok_all = ok_all and st["rc"] == 0 and c.get("replicas_bit_identical") is True
    # 4. the curve
    values = {}
    for config, extra in (("cfg4", ["--scaling", "strong"]),) if a.quick else (("cfg3", []), ("cfg4", ["--scaling", "strong"]), ("cfg5", [])):
        for collective in ("rccl", "auto"):
            for n in worlds:
                if n == 1 and collective == "auto":
                    continue
                st, d = bench(n, config, collective, extra)
                st["what"] = "%s --gpus %d --collective %s" % (config, n, collective)
                report["steps"].append(st)
                if d:
                    values.setdefault("%s/%s" % (config, collective if n > 1 else "single"), {})[str(n)] = d["value"]
                    if n > 1:
                        ok_all = ok_all and d.get("collectives", {}).get("replicas_bit_identical") is True
                ok_all = ok_all and st["rc"] == 0
    report["values"] = values
    # 5. the C++ driver, N ranks, replicas' checkpoints byte for byte
    if N > 1:
        tmp = tempfile.mkdtemp(prefix="first8_")
        exe = os.path.join(ROOT, "ppo_cpp_amd", "ppo_cpp_hip")
        cmd = [exe, "--ranks", str(N), "--threads", str(16 * N), "--batch_steps", "32", "--hidden", "64,64", "--epochs", "2", "--minibatches", "4",
               "--steps", str(4 * 16 * N * 32), "--seeded", "--saves", "1", "--dir", tmp, "--id", "run", "--replica_saves"]
        if a.stand_in:
            cmd += ["--devices", "0"]
        st, out = child(cmd, env)
        st["what"] = "ppo_cpp_hip --ranks %d" % N
        digests = {}
        for r in range(N):
            h = hashlib.sha256()
            for ext in (".index", ".data-00000-of-00001"):
                fn = os.path.join(tmp, ("run.pkl.0" if r == 0 else "run.pkl.rank%d.0" % r) + ext)
                h.update(open(fn, "rb").read() if os.path.exists(fn) else b"missing:" + fn.encode())
            digests[str(r)] = h.hexdigest()[:16]
        st["checkpoint_digests"] = digests
        report["cpp_driver_replicas_byte_identical"] = st["rc"] == 0 and len(set(digests.values())) == 1
        ok_all = ok_all and report["cpp_driver_replicas_byte_identical"]
        report["steps"].append(st)
    report["replicas_bit_identical"] = ok_all
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report.get(k) for k in ("devices", "rccl_nranks", "distinct_devices", "graph_collectives_rccl", "peer_path_used",
                                                 "cpp_driver_replicas_byte_identical", "replicas_bit_identical", "values")}))
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
