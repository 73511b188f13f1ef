"""gys_regex.hpp (the `like` criteria's linear-time matcher) on the CPU: limits of ADVICE r5 (nested counted repetitions, counts above 1000,
literal braces) and the accepted subset against std::regex -- tests/cpp/regex/test_regex.cc, plain g++."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_regex_limits_and_subset():
    exe = os.path.join(tempfile.gettempdir(), "gys_test_regex")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "regex", "test_regex.cc"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "regex ok" in r.stdout
