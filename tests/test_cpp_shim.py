"""The C++17 host-side mirror of MCONN_HANDLER's entry points (gys_mconn_shim.hpp) compiles with plain g++ against the C ABI
(CPU check) and produces the reference's LISTEN_SUMM_STATS / STATE_ONE results on the GPU (gpu check)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from gyeeta_amd import build
    if build.needs_build():
        build.build()
    exe = str(tmp_path / "test_shim")
    lib = os.path.join(ROOT, "gyeeta_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_shim.cc"), "-o", exe,
                           "-L" + lib, "-lgysketch", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return exe


def test_shim_compiles_and_links_with_gxx(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.check_output([exe]).decode()
    assert "link ok, abi 4" in out


@pytest.mark.gpu
def test_shim_runs_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run(["timeout", "-s", "KILL", "120", exe, "run"], capture_output=True, text=True)
    assert r.returncode == 0 and "shim ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_shim_window_close_rccl_one_rank(tmp_path):
    """the window boundary with the collectives inside the library (gys_window_close_rccl) from plain C++ with a one-rank communicator.
    RCCL's own bootstrap (ncclCommInitRank) does not return on part of the GPU pool; that is reported as a skip, not as a hang."""
    exe = _build(tmp_path)
    r = subprocess.run(["timeout", "-s", "KILL", "60", exe, "rccl"], capture_output=True, text=True)
    if r.returncode == -9 and "[shim] rccl join" in r.stderr and "[shim] rccl joined" not in r.stderr:
        pytest.skip("ncclCommInitRank did not return within 60 s on this box (RCCL bootstrap); the in-library exchange was not exercised")
    assert r.returncode == 0 and "shim rccl ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
