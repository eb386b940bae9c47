"""tools/ holds GPU-box diagnostics; this keeps them from rotting: every script must at least parse, and the ones the documents name must exist."""
import glob
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_tool_parses():
    py = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    sh = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert py and sh
    for f in py:
        py_compile.compile(f, doraise=True)
    for f in sh:
        assert subprocess.run(["bash", "-n", f]).returncode == 0, f


def test_tools_named_in_the_documents_exist():
    text = open(os.path.join(ROOT, "DESIGN.md")).read() + open(os.path.join(ROOT, "tools", "README.md")).read()
    for name in set(re.findall(r"tools/([A-Za-z0-9_]+\.(?:py|sh))", text)):
        assert os.path.exists(os.path.join(ROOT, "tools", name)), name


import pytest


@pytest.mark.gpu
def test_first_multi_device_command_dry_run_on_one_gpu(tmp_path):
    """tools/first_8gpu.sh, the ONE command for the first run on more than one device, as a dry run on the one test GPU (--stand-in: every rank on device 0, the
    shared-memory stand-in behind the nccl* calls; --quick: cfg4 only, 2 steps): both exchange paths at world 4, the curve's bench lines, `ppo_cpp_hip --ranks 4`, and
    the one JSON it writes -- so that the day a node exists the command is known to run."""
    import json
    from tests.test_dp_two_ranks import build_fake_rccl
    fake = build_fake_rccl(str(tmp_path))
    out = os.path.join(str(tmp_path), "first.json")
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "first_8gpu.sh"), "--out", out, "--max-gpus", "4", "--quick", "--stand-in", fake],
                       capture_output=True, text=True, timeout=1500)
    rep = json.load(open(out))
    assert p.returncode == 0, json.dumps([{k: s[k] for k in ("what", "rc", "stderr_tail")} for s in rep["steps"] if s["rc"] != 0])[:3000]
    assert rep["rccl_nranks"] == 4 and rep["distinct_devices"] == 1 and rep["replicas_bit_identical"] is True and rep["cpp_driver_replicas_byte_identical"] is True
    assert rep["peer_path_used"] is True
    assert set(rep["values"]["cfg4/rccl"]) == {"2", "4"} and set(rep["values"]["cfg4/single"]) == {"1"} and all(v > 0 for v in rep["values"]["cfg4/auto"].values())


def test_the_pytest_tail_carries_the_finding_of_a_failed_and_an_xfailed_test(tmp_path):
    """tests/conftest.py's terminal summary: one line per failed / xfailed test with the head of its assertion message, at the very END of the output (`pytest -q` prints nothing
    for an xfail, and round 5's finding was lost in exactly that way).  A throw-away test file run under this repository's conftest."""
    import sys
    import shutil
    d = tmp_path / "tests"
    d.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "conftest.py"), str(d / "conftest.py"))
    (d / "test_x.py").write_text(
        "import pytest\n"
        "def test_plain():\n    assert 1 + 1 == 3, 'first differing buffer: thetaT'\n"
        "@pytest.mark.xfail(strict=False, reason='open')\n"
        "def test_expected():\n    raise AssertionError('handle 1 differs first at iteration 1: loss rows')\n")
    p = subprocess.run([sys.executable, "-m", "pytest", str(d), "-q", "-p", "no:cacheprovider"], capture_output=True, text=True, cwd=str(tmp_path))
    tail = p.stdout.strip().splitlines()[-12:]
    text = "\n".join(tail)
    assert "findings (tests/conftest.py)" in text
    assert any(l.startswith("FAILED") and "first differing buffer: thetaT" in l for l in tail), text
    assert any(l.startswith("XFAILED") and "handle 1 differs first at iteration 1: loss rows" in l for l in tail), text
