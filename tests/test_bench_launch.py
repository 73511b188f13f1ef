"""bench.py's N > 1 entry: `python bench.py --gpus N` with no launcher around it starts its own ranks; the cross-rank
checksum exchange that ends an N-GPU run is exercised here over gloo with two CPU ranks (--selftest-launch); a box with fewer devices
than ranks gets a clear message, not a launcher traceback."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_without_devices_fails_cleanly():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two devices")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2
    assert "needs 2 devices" in r.stderr and "Traceback" not in r.stderr


def test_self_launch_two_ranks_checksums_agree():
    r = _run("--gpus", "2", "--selftest-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["selftest"] and d["n_gpus"] == 2
    assert d["exchange_check"]["ranks_seen"] == 2 and d["exchange_check"]["ranks_consistent"] is True


def test_self_launch_detects_a_rank_that_disagrees():
    r = _run("--gpus", "2", "--selftest-launch", "--selftest-corrupt-rank", "1")
    assert r.returncode != 0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["exchange_check"]["ranks_consistent"] is False


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_whole_flow():
    """`python bench.py --gpus 2 --share-device`: the whole N > 1 flow of the bench on a one-GPU box -- self-launch, host-hash sharding,
    communicator join through the C ABI, gys_window_close_rccl in every window (RCCL entry points served by tests/cpp/fakerccl, both ranks
    on device 0), the cross-rank register check and the side configuration compared with a single-rank engine"""
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="gys_bench_detail_"), "detail.json")
    r = _run("--gpus", "2", "--share-device", "--hosts", "400", "--svcs", "100", "--events", str(1 << 22), "--steps", "4", "--warmup", "1",
             "--prime-windows", "2", "--no-cpu-baseline", "--no-host-fed", "--strict-exchange", "--detail-out", detail, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 4096  # the compact line (what the driver parses); the full result is the detail file
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "rccl_in_library" and d["config"]["exchange_fallback"] is False
    assert d["exchange_check"]["ranks_seen"] == 2 and d["exchange_check"]["ranks_consistent"] is True and d["exchange_check"]["ok"] is True
    assert d["parity_ok"] is True and d["value"] > 0
    x = json.load(open(detail))["exchange_check"]
    assert x["ranks_seen"] == 2 and x["ranks_consistent"] is True and x["ok"] is True
    assert x["side_config"]["ranks_consistent"] is True and x["side_config"]["equals_single_rank_engine"] is True
