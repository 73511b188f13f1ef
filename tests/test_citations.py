"""every `file:line` citation of the reference in this repo's sources and docs names an existing reference file and lines inside it
(tools/check_citations.py; only where /root/reference is mounted, i.e. in the build container)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_citations_resolve():
    if not os.path.isdir(os.environ.get("GY_REFERENCE_DIR", "/root/reference")):
        pytest.skip("no reference tree here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_citations.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 problems" in r.stdout and int(r.stdout.strip().split("\n")[-1].split()[0]) > 200
