"""The bench.py output contract (one JSON line on stdout with the agreed keys), checked on a short real run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 1e5 and abs(d["value"] - 65536.0 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "rocprof_avg_us", "rocprof_source"):
        assert k in r, k
    if r["traffic"] is not None:
        assert "profiles/" in r["traffic_source"] and "offline" in r["traffic_source"]
    if r["rocprof_avg_us"] is not None:
        assert "profiles/" in r["rocprof_source"] and 0.2 < r["rocprof_avg_us"] / r["avg_us"] < 1.5     # the event timing is an upper bound
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    # the stated baseline is the vectorised port on ONE thread over a WHOLE update; the same port on all granted cores and the scalar
    # C restatement stay on record beside it
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["covers"] == 1.0 and "Adam" in c["covers_what"]
    a = c["all_cores"]
    assert a["cores"] >= 1 and a["value"] >= 0.8 * c["value"] and a["covers"] == 1.0         # more threads must not come out slower than one
    assert c["scalar_port"]["cores"] == 1 and 0 < c["scalar_port"]["value"] < c["value"]


def test_cpu_baseline_leg_runs_without_a_gpu_and_covers_a_whole_update():
    """CPU-runnable: the vectorised leg in its child process (BLAS threads fixed before NumPy loads) on a small configuration"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.cpu_vectorised_leg("cfg4", 1, 1.0)
    assert r["value"] > 0 and r["cores"] == 1 and r["covers"] == 1.0 and "train steps" in r["sample"]
    assert bench.granted_cores() >= 1


@pytest.mark.gpu
def test_bench_roofline_of_a_multi_kernel_class_is_consistent_at_cfg5():
    """bf16 path: the timed class `train_fwd_bwd` is a SEQUENCE of launches per train step; the profile-derived numbers in the roofline
    block (rocprof_avg_us, traffic) must describe the same sequence as flop_per_launch / avg_us, not one of its kernels."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    r = d["roofline"]
    assert d["dtype"] == "bf16" and r["kernel"] == "train_fwd_bwd" and r["peak"] == 2500.0
    if r["rocprof_avg_us"] is not None:
        assert 0.5 < r["rocprof_avg_us"] / r["avg_us"] < 1.5 and "1 x gemm_chain_bf16_kernel<0>" in r["rocprof_source"] and "PPO_HIP_NO_GRAPH" in r["rocprof_source"]
    if r["traffic"] is not None:
        # arithmetic intensity of the class from the two profile-derived numbers: bf16 GEMMs of K = 1024 at 256 x 128 tiles sit at a few
        # hundred FLOP per HBM byte; one kernel's traffic under the whole class's FLOP (the round-3 mix-up) gave > 1300
        assert 50.0 < r["flop_per_launch"] / r["traffic"] < 800.0 and "1 x gemm_chain_bf16_kernel<1>" in r["traffic_source"]
