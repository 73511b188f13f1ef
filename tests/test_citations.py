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


def test_environment_switches_are_documented():
    """every environment variable the library or its ctypes stub reads appears in INTEGRATION.md's table (section 9)"""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "gyeeta_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "gyeeta_amd", "*.py")):
        text = open(path, errors="replace").read()
        names.update(re.findall(r'getenv\("(GYS_[A-Z0-9_]+)"\)', text))
        names.update(re.findall(r'environ(?:\.get)?[\[(]"(GYS_[A-Z0-9_]+)"', text))
    names.discard("GYS_DBG")  # read only by -DGYS_RESP_DBG=1 timing builds (tools/ab_libs.sh)
    assert len(names) >= 8, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
