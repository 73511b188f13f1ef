"""Oracle (oracle/gy_oracle.c) against every known-answer vector the reference holds for this path:
test/test_histogram.cc:28-83 (Hist_9_26), :98-147 (Hist_n4), the SURVEY 8c hash KATs and probe outputs, and the
committed golden fixtures under tests/golden/ (generated from the reference's own code by tests/golden/make_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hist(o, kind):
    h = o.Hist()
    o.lib().gyo_hist_init(C.byref(h), kind)
    return h


def _pct(o, h, p):
    pd = (o.HistData * 1)()
    pd[0].percentile = p
    o.lib().gyo_hist_percentiles(C.byref(h), pd, 1, None, None, None)
    return pd[0].data_value


def test_hist_9_26_reference_asserts(oracle):
    # test/test_histogram.cc:28-83
    L = oracle.lib()
    k = oracle.KINDS["FIXED_9_26_5"]
    assert L.gyo_hist_nbuckets(k) == 6
    h = _hist(oracle, k)
    seq = [(0, 0), (8, 0), (9, 1), (10, 1), (13, 1), (14, 2), (15, 2), (18, 2), (19, 3), (20, 3), (23, 3), (24, 4)]
    for v, b in seq:
        assert L.gyo_hist_add(C.byref(h), v) == b
    assert _pct(oracle, h, 75.0) == 23
    for v, b in [(25, 4), (26, 4), (27, 5), (40, 5)]:
        assert L.gyo_hist_add(C.byref(h), v) == b
    assert _pct(oracle, h, 90.0) == 26


def test_hist_n4_reference_asserts(oracle):
    # test/test_histogram.cc:98-147
    L = oracle.lib()
    k = oracle.KINDS["FIXED_N15_N3_4"]
    assert L.gyo_hist_nbuckets(k) == 6
    h = _hist(oracle, k)
    for v, b in [(0, 5), (-16, 0), (-15, 1), (-13, 1), (-12, 1), (-11, 2), (-10, 2), (-8, 2)]:
        assert L.gyo_hist_add(C.byref(h), v) == b
    assert _pct(oracle, h, 75.0) == -8
    for v, b in [(-7, 3), (-5, 3), (-4, 3), (-3, 4), (-2, 5)]:
        assert L.gyo_hist_add(C.byref(h), v) == b
    assert _pct(oracle, h, 75.0) == -4
    assert L.gyo_hist_add(C.byref(h), 2) == 5


def test_resp_hist_survey_probe(oracle):
    # test/test_histogram.cc:154-170 stream; expected output recorded by the survey's run of the unmodified test
    L = oracle.lib()
    h = _hist(oracle, 0)
    for v in [0, 2, 2000, 1000000] + list(range(1000)):
        L.gyo_hist_add(C.byref(h), v)
    counts = [h.stats[i].count for i in range(15)]
    assert counts == [0, 3, 10, 20, 30, 40, 50, 50, 100, 150, 250, 299, 1, 0, 1]
    assert [_pct(oracle, h, p) for p in (25.0, 50.0, 75.0, 95.0, 99.0, 99.99)] == [300, 700, 1000, 1000, 1000, 3000]


def test_hash_kats(oracle):
    # SURVEY 8c [probe] KATs from the reference headers
    L = oracle.lib()
    assert L.gyo_get_uint64_hash(1) == 0x64E92BD9
    ip, v6 = oracle.ip_bytes(0x0100007F)
    assert L.gyo_ip_port_hash(ip, v6, 8080, 0) == 0x44B2FA5B
    ip2, _ = oracle.ip_bytes(0x0200007F)
    assert L.gyo_pair_ip_port_hash(ip, 0, 8080, ip2, 0, 80) == 0xA0EEE7A1  # recorded from oracle/_ref in this container


def test_golden_fixture(oracle):
    path = os.path.join(GOLD, "ref_vectors.json")
    if not os.path.exists(path):
        pytest.skip("golden fixture missing")
    g = json.load(open(path))
    L = oracle.lib()
    for rec in g["jhash2"]:
        w = np.array(rec["words"], dtype=np.uint32)
        assert L.gyo_jhash2(oracle.ptr(w, oracle.u32p), len(w), rec["seed"]) == rec["hash"]
    for rec in g["jhash_bytes"]:
        b = bytes.fromhex(rec["hex"])
        assert L.gyo_jhash(b, len(b), rec["seed"]) == rec["hash"]
    for rec in g["uint64_hash"]:
        assert L.gyo_get_uint64_hash(rec["key"]) == rec["hash"]
    for rec in g["machine_id"]:
        assert L.gyo_machine_id_hash(rec["first"], rec["second"]) == rec["hash"]
    for rec in g["ip_port"]:
        ip, v6 = oracle.ip_bytes(bytes.fromhex(rec["ip"]) if rec["v6"] else rec["ip"])
        assert L.gyo_ip_port_hash(ip, v6, rec["port"], rec["ignore_ip"]) == rec["hash"]
        assert L.gyo_ns_ip_port_hash(ip, v6, rec["port"], rec["inode"], rec["ignore_ip"]) == rec["ns_hash"]
    for rec in g["pair_ip_port"]:
        cip, c6 = oracle.ip_bytes(bytes.fromhex(rec["cip"]) if rec["c6"] else rec["cip"])
        sip, s6 = oracle.ip_bytes(bytes.fromhex(rec["sip"]) if rec["s6"] else rec["sip"])
        assert L.gyo_pair_ip_port_hash(cip, c6, rec["cport"], sip, s6, rec["sport"]) == rec["hash"]
    for rec in g["hist"]:
        kind = rec["kind"]
        h = _hist(oracle, kind)
        vals = np.array(rec["values"], dtype=np.int64)
        L.gyo_hist_add_many(C.byref(h), oracle.ptr(vals, oracle.i64p), len(vals))
        nb = L.gyo_hist_nbuckets(kind)
        assert nb == rec["nbuckets"]
        assert [h.stats[i].count for i in range(nb)] == rec["counts"]
        assert [h.stats[i].sum for i in range(nb)] == rec["sums"]
        assert h.total_count == rec["total"] and h.max_val_seen == rec["max"]
        pd = (oracle.HistData * len(rec["pcts"]))()
        for i, p in enumerate(rec["pcts"]):
            pd[i].percentile = p
        avg = C.c_float()
        L.gyo_hist_percentiles(C.byref(h), pd, len(rec["pcts"]), None, None, C.byref(avg))
        assert [d.data_value for d in pd] == rec["pct_values"]
        assert [d.sum for d in pd] == rec["pct_sums"]
        assert [d.count for d in pd] == rec["pct_counts"]
        assert np.float32(avg.value).tobytes() == np.float32(rec["avg"]).tobytes()
        assert [L.gyo_bucket_max_threshold(kind, i) for i in range(nb + 1)] == rec["thresholds"]
    for rec in g["topn"]:
        v = np.array(rec["values"], dtype=np.uint64)
        out = np.zeros(rec["n"], dtype=np.uint64)
        k = L.gyo_topn_u64(oracle.ptr(v, oracle.u64p), len(v), rec["n"], oracle.ptr(out, oracle.u64p))
        assert out[:k].tolist() == rec["top"]
    # time levels: bucket numbering and percentile rule of the reference's in-tree slab-histogram container
    assert g["slab"]["nbuckets"] == L.gyo_hist_nbuckets(oracle.RESP_TIME_HASH)
    for rec in g["slab"]["bucket_idx"]:
        assert L.gyo_bucket(oracle.RESP_TIME_HASH, rec["value"]) == rec["idx"]
    for rec in g["slab"]["percentile_idx"]:
        c = np.array(rec["counts"], dtype=np.uint64)
        assert L.gyo_slab_percentile_idx(oracle.ptr(c, oracle.u64p), len(c), rec["pct"]) == rec["idx"]
