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
