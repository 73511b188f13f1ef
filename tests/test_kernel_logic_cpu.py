"""Kernel LOGIC without a GPU: gyeeta_amd/csrc/gys_kernels.hpp compiled by g++ against a small CPU stand-in of the HIP device model
(tests/cpp/kemu/hip/hip_runtime.h: one OS thread per GPU thread, barriers for __syncthreads and the wave64 exchanges, host atomics) and
the hot kernels run on synthetic input:
  * k_digest_bins on hand-made keys (tests/cpp/kemu/test_bins.cc): re-clustered digests, lazily folded histogram records, CONN_BITMAP
    rows, min / max and drained meta records equal the oracle's;
  * the response-event pipeline k_resp_host (+ finalize_key) -> k_digest_bins / k_digest_merge over several batches and window
    boundaries (tests/cpp/kemu/test_resp.cc): counters, HLL registers, all-service histogram, every key's buffered values and digest and
    the records of re-clustered keys equal the oracle's sequential engine fed the same bytes.
  * the paths of a key whose batch does not fit its buffer (tests/cpp/kemu/test_spill.cc): spill in finalize_key, the SPILL pass of
    k_resp_host, merges from buffer + run in every size class, the several-workgroup path of gys_huge.hpp with its sorted tail and its
    one-workgroup fallback.
This does not replace the -m gpu parity tests (no memory model, no execution masks, no timing): it catches logic errors in kernel
changes before GPU minutes are spent on them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEMU = os.path.join(ROOT, "tests", "cpp", "kemu")


def _build(tmp_path_factory, oracle, src, name, defs=()):
    oracle.lib()  # builds oracle/liboracle.so if needed
    exe = str(tmp_path_factory.mktemp("kemu") / name)
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-w", "-I" + KEMU] + ["-D" + d for d in defs] +
                          [os.path.join(KEMU, src), "-o", exe, "-L" + odir, "-l:liboracle.so", "-Wl,-rpath," + odir, "-pthread"])
    return exe


@pytest.fixture(scope="module")
def kemu_bins(tmp_path_factory, oracle):
    return _build(tmp_path_factory, oracle, "test_bins.cc", "kemu_bins")


@pytest.mark.parametrize("seed", [12345, 7, 99])
def test_digest_bins_kernel_logic_equals_oracle(kemu_bins, seed):
    r = subprocess.run(["timeout", "-s", "KILL", "300", kemu_bins, str(seed)], capture_output=True, text=True)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip())
    assert r.returncode == 0 and "kemu bins ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("tpt,split", [(16, False), (12, False), (16, True)], ids=["tiles-16384", "tiles-6144", "split-form"])
def test_resp_pipeline_kernel_logic_equals_oracle_engine(tmp_path_factory, oracle, tpt, split):
    """split-form: long segments cut into parts of 65 536 events, several workgroups per host reserving buffer space with device
    atomics, k_key_finalize as its own launch"""
    exe = _build(tmp_path_factory, oracle, "test_resp.cc", "kemu_resp%d%s" % (tpt, "s" if split else ""),
                 ["KEMU_TPT=%d" % tpt] + (["KEMU_SPLIT", "KEMU_NB=3"] if split else []))
    r = subprocess.run(["timeout", "-s", "KILL", "900", exe, "4242"], capture_output=True, text=True)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip())
    assert r.returncode == 0 and "kemu resp ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_spill_and_huge_paths_kernel_logic_equals_oracle_engine(tmp_path_factory, oracle):
    exe = _build(tmp_path_factory, oracle, "test_spill.cc", "kemu_spill")
    r = subprocess.run(["timeout", "-s", "KILL", "900", exe, "777"], capture_output=True, text=True)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip())
    assert r.returncode == 0 and "kemu spill ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
