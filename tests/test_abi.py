"""CPU-side checks of the drop-in boundary: libgysketch.so loads without a GPU, exports every symbol include/gysketch.h declares,
the ctypes table covers exactly that set, struct sizes agree, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "gysketch.h")


@pytest.fixture(scope="module")
def capi():
    from gyeeta_amd import build, capi as c
    if build.needs_build():
        build.build()
    c.load()
    return c


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gys_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(capi):
    names = declared_functions()
    assert len(names) > 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (gys_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in gysketch.h but not exported: {missing}"
    assert sorted(capi.SIGNATURES) == names, (set(capi.SIGNATURES) ^ set(names))


def test_struct_sizes_match_header(capi, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "gysketch.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(gys_config),sizeof(gys_listener_info),sizeof(gys_resp_seg),sizeof(gys_host_state),sizeof(gys_reduce_section),"
                   "sizeof(gys_svcsumm),sizeof(gys_cluster_state),sizeof(gys_hist_data),sizeof(gys_hist_rec),sizeof(gys_topn_entry),sizeof(gys_counters),sizeof(gys_time_hist_val),sizeof(gys_listener_day_stats));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])  # header is plain C
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(t) for t in (capi.Config, capi.ListenerInfo, capi.RespSeg, capi.HostState, capi.ReduceSection, capi.SvcSumm,
                                 capi.ClusterState, capi.HistData, capi.HistRec, capi.TopnEntry, capi.Counters, capi.TimeHistVal,
                                 capi.ListenerDayStats)]
    assert got == exp
    assert C.sizeof(capi.HistRec) == 256 and C.sizeof(capi.SvcSumm) == 52 and C.sizeof(capi.ClusterState) == 44
    assert C.sizeof(capi.ListenerDayStats) == 48  # comm::LISTENER_DAY_STATS


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = capi.load()
    cfg = capi.Config()
    cfg.struct_size = C.sizeof(capi.Config)
    cfg.device, cfg.nranks, cfg.max_hosts, cfg.max_services, cfg.max_clusters = -1, 1, 4, 16, 2
    h = C.c_void_p()
    rc = L.gys_create(C.byref(cfg), C.byref(h))
    assert rc == capi.ERR_HIP and b"hipGetDeviceCount" in L.gys_last_error()
    from gyeeta_amd.engine import SketchEngine
    with pytest.raises(RuntimeError):
        SketchEngine(max_hosts=1, max_services=1)


def test_bad_config_rejected(capi):
    L = capi.load()
    cfg = capi.Config()
    h = C.c_void_p()
    assert L.gys_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVAL  # struct_size 0
    assert L.gys_abi_version() == 7


def test_shard_function_is_reference_machine_id_hash(capi, oracle):
    from gyeeta_amd import wire
    from gyeeta_amd.engine import mid_buf
    L = capi.load()
    for h in range(200):
        mid = wire.machine_id(h)
        first, second = int.from_bytes(mid[:8], "little"), int.from_bytes(mid[8:], "little")
        ref = oracle.lib().gyo_machine_id_hash(first, second)  # GY_MACHINE_ID::get_hash (pinned against oracle/_ref)
        assert L.gys_machine_id_hash(mid_buf(mid)) == ref
        for n in (1, 2, 4, 8):
            assert L.gys_shard_of(mid_buf(mid), n) == ref % n


def test_build_stamp_names_sources_and_device_code():
    """gyeeta_amd/lib/build_commit.txt = "<commit>[+dirty] <sha256-16 of the source files> <sha256-16 of the gfx950 code object's .rodata + .text>":
    the device-code hash is what bench.py compares with profiles/pmc_traffic.json (roofline.traffic.same_kernels_as_this_run); it is a
    property of the built library alone (recomputed here from the .so) and does not move when the stamp is rewritten"""
    import re
    from gyeeta_amd import build
    build.build()
    if build.build_commit() is None:
        build.stamp_commit()
    src, dev = build.sources_sha(), build.device_code_sha()
    assert src and re.fullmatch(r"[0-9a-f]{16}", src), src
    assert dev and re.fullmatch(r"[0-9a-f]{16}", dev), dev
    assert build._hash_device_code(build.LIB_PATH) == dev
    assert src == build._hash_sources()


def test_bench_compares_counter_evidence_by_device_code():
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    dev = b._device_code()
    assert dev
    assert b._same_kernels({"device_code": dev, "source_kernels": "0" * 16}) is True      # same kernels, sources moved (host-only change)
    assert b._same_kernels({"device_code": "f" * 16, "source_kernels": b._kernel_sources()}) is False
    assert b._same_kernels({"source_kernels": b._kernel_sources()}) is True                # older evidence files: source files only
    assert b._same_kernels({}) is None
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert isinstance(b._same_kernels(t), bool)
