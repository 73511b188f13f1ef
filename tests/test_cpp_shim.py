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
    assert "link ok, abi 7" in out


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
    # Two RCCL builds are on the box: /opt/rocm/lib/librccl.so (ROCm 7.2; the library's rpath) and the one PyTorch bundles (RCCL 2.26.6,
    # ROCm 7.0 -- the generation of the pool's host driver).  On part of the pool the 7.2 build never returns from ncclCommInitRank (a
    # one-rank communicator, loopback bootstrap) while the bundled one does, so the bundled build is tried first; same soname, same API.
    envs = []
    try:
        import torch
        envs.append(dict(os.environ, LD_LIBRARY_PATH=os.path.join(os.path.dirname(torch.__file__), "lib")))
    except Exception:
        pass
    envs.append(dict(os.environ))
    down = []
    for k, env in enumerate(envs):
        # (a communicator comes up in 1 - 3 s where it comes up at all; a box on which the first build hangs rarely answers the second)
        r = subprocess.run(["timeout", "-s", "KILL", "45" if k == 0 else "30", exe, "rccl"], capture_output=True, text=True, env=env)
        # RCCL's own bootstrap failing on this box (ncclGetUniqueId / ncclCommInitRank never returning, or returning an error: exit codes
        # 18 / 19 of tests/cpp/test_shim.cc, both before "[shim] rccl joined") is the box's, not the library's: next build, else skip.
        # Everything before the bootstrap (the same steps as the `run` mode) and everything after it must pass.
        in_rccl = "[shim] rccl\n" in r.stderr
        at_bootstrap = in_rccl and "[shim] rccl joined" not in r.stderr
        if (at_bootstrap and r.returncode in (18, 19)) or (in_rccl and r.returncode == -9):  # (-9: a collective of this RCCL build never returned)
            down.append((r.returncode, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""))
            continue
        assert r.returncode == 0 and "shim rccl ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
        return
    pytest.skip(f"RCCL's bootstrap did not come up with any RCCL build on this box {down}; the in-library exchange was not exercised")
