import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run the iteration-heavy soak cases (marked slow); tools/soak.py passes it")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: an iteration-heavy soak case of a body that has a short case in -m gpu; skipped unless --runslow (the driver's -m gpu run has 1200 s)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="soak case: run with --runslow (tools/soak.py)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line per failed / xfailed test with what its assertion FOUND (the head of the message: the first differing buffer, the first differing output), printed at the very
    end so that a driver that keeps only the tail of the log still carries the finding (`pytest -q` prints nothing at all for an xfail)."""
    lines = []
    for kind in ("failed", "error", "xfailed"):
        for rep in terminalreporter.stats.get(kind, []):
            msg = ""
            crash = getattr(getattr(rep, "longrepr", None), "reprcrash", None)
            if crash is not None:
                msg = crash.message
            elif rep.longrepr is not None:
                msg = str(rep.longrepr)
            head = [l.strip() for l in msg.splitlines() if l.strip()][:3]
            lines.append("%s %s: %s" % (kind.upper(), rep.nodeid, " | ".join(head)[:600] or getattr(rep, "wasxfail", "")))
    if lines:
        terminalreporter.write_sep("=", "findings (tests/conftest.py)")
        for l in lines:
            terminalreporter.write_line(l)
    # which HIP runtime served this process (the torch wheel bundles one of its own under the system's soname; whichever is loaded first serves everything)
    try:
        with open("/proc/self/maps") as f:
            rt = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
        if rt:
            terminalreporter.write_line("HIP runtime serving this process: " + ", ".join(rt))
    except OSError:
        pass
