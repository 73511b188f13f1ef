"""The LOGIC of the hot t-digest merge kernel without a GPU: gyeeta_amd/csrc/gys_kernels.hpp compiled by g++ against a small CPU stand-in
of the HIP device model (tests/cpp/kemu/hip/hip_runtime.h: one OS thread per GPU thread, barriers for __syncthreads and the wave64
exchanges) and k_digest_bins run on synthetic keys; the re-clustered digests, the lazily folded histogram records, CONN_BITMAP rows,
min / max and the drained meta records must equal the oracle's.  This does not replace the -m gpu parity tests (no memory model, no
execution masks, no timing): it catches logic errors in kernel changes before GPU minutes are spent on them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kemu_bins(tmp_path_factory, oracle):
    oracle.lib()  # builds oracle/liboracle.so if needed
    exe = str(tmp_path_factory.mktemp("kemu") / "kemu_bins")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "tests", "cpp", "kemu"),
                           os.path.join(ROOT, "tests", "cpp", "kemu", "test_bins.cc"), "-o", exe, "-L" + odir, "-l:liboracle.so",
                           "-Wl,-rpath," + odir, "-pthread"])
    return exe


@pytest.mark.parametrize("seed", [12345, 7, 99])
def test_digest_bins_kernel_logic_equals_oracle(kemu_bins, seed):
    r = subprocess.run(["timeout", "-s", "KILL", "300", kemu_bins, str(seed)], capture_output=True, text=True)
    assert r.returncode == 0 and "kemu bins ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
