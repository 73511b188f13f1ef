"""CPU checks of the per-listener scan restatement (oracle/gy_oracle_lscan.c): the pieces that the reference's own headers can pin
(get_bucketid_from_threshold, oracle/_ref) and the restatement's arithmetic against a second, independent numpy formulation."""
import ctypes as C

import numpy as np


def test_bucketid_from_threshold_equals_reference(oracle, reflib):
    L = oracle.lib()
    vals = list(range(-5, 40)) + [60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000, 15001, 2**31 - 1, 99999, 59, 61]
    for v in vals:
        assert reflib.ref_resp_bucketid_from_threshold(v) == L.gyo_bucketid_from_threshold(oracle.RESP_TIME_HASH, v), v


def test_listener_scan_one_against_numpy(oracle):
    """three windows into a fresh multi-level histogram: the 5-s level is the last window, the others everything so far; counts, sums,
    percentile ceilings (first bucket whose cumulative fraction reaches p), bucket ids, QPS arithmetic and the CONN_BITMAP breakup"""
    L = oracle.lib()
    rng = np.random.default_rng(4)
    thr = [1, 10, 30, 60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000]
    h = oracle.MLHist()
    L.gyo_mlh_init(C.byref(h), oracle.RESP_TIME_HASH, 10)
    qps, act = oracle.Hist(), oracle.Hist()
    L.gyo_hist_init(C.byref(qps), oracle.KINDS["SEMI_LOG_HASH_LO"])
    L.gyo_hist_init(C.byref(act), oracle.KINDS["HASH_1_3000"])
    for v in (12, 40, 40, 900, 3):
        L.gyo_hist_add(C.byref(qps), v)
        L.gyo_hist_add(C.byref(act), v % 7)
    tot = np.zeros((15, 2), dtype=np.int64)
    t = 1_700_000_000
    last = None
    for w in range(3):
        t += 5
        vals = np.minimum(np.floor(rng.lognormal(2.0 + w, 1.2, 500)), 20000).astype(np.int64)
        b = np.array([L.gyo_bucket(oracle.RESP_TIME_HASH, int(v)) for v in vals])
        st = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
        for k in range(15):
            st["count"][k] = int((b == k).sum())
            st["sum"][k] = int(vals[b == k].sum())
        L.gyo_mlh_add_hist(C.byref(h), t, st.ctypes.data, 1)
        L.gyo_mlh_flush(C.byref(h), t)
        last = np.stack([st["count"][:15].astype(np.int64), st["sum"][:15]], axis=1)
        tot += last
    rows = np.zeros(64, dtype=np.uint16)  # 32 rows of resp_bitmap_v4_, 32 rows of resp_bitmap_v6_
    for port, bucket in ((1000, 3), (1001, 3), (1033, 3), (77, 9), (78, 9), (5, 0)):
        L.gyo_conn_bitmap_add(oracle.ptr(rows, oracle.u16p), port, bucket)
    rows6 = rows[32:]
    for port, bucket in ((1000, 3), (40, 9), (41, 12)):  # IPv6 clients: their own bitmap, counts added per bucket (gy_socket_stat.cc:4144-4149)
        L.gyo_conn_bitmap_add(oracle.ptr(rows6, oracle.u16p), port, bucket)
    notify = np.zeros(88, dtype=np.uint8)
    out = oracle.ListenerScan()
    L.gyo_listener_scan_one(C.byref(h), C.byref(qps), C.byref(act), oracle.ptr(rows, oracle.u16p), 0xabcdef0123, 2.5, 5, oracle.ptr(notify, oracle.u8p), C.byref(out))

    def pct(c, p):
        if c.sum() == 0:
            return thr[0]
        cum = np.cumsum(c) / c.sum()
        for i in range(15):
            if c[i] and p / 100.0 <= cum[i]:
                return ([-1] + thr + [2**31 - 1])[i] if i < 14 else 2**31 - 1
        return 2**31 - 1

    for lv, rec in ((0, last), (1, tot), (2, tot), (3, tot)):
        assert out.tcount[lv] == rec[:, 0].sum() and out.tsum[lv] == rec[:, 1].sum()
        assert out.p95_ms[lv] == max(0, pct(rec[:, 0], 95.0)) and out.p99_ms[lv] == max(0, pct(rec[:, 0], 99.0)) and out.p25_ms[lv] == max(0, pct(rec[:, 0], 25.0))
    nq = int(last[:, 0].sum())
    assert out.last_qps == int(np.float32(np.float32(nq) * np.float32(2.5)) / np.float32(5.0))
    assert out.curr_qps == max(out.last_qps, nq // 5)
    assert out.b5 == (thr.index(out.p95_ms[0]) + 1 if out.p95_ms[0] in thr else 14)
    assert list(out.nactive_conn_arr) == [1, 0, 0, 3, 0, 0, 0, 0, 0, 3, 0, 0, 1, 0, 0] and out.nconn_active == 3
    # (ports 1001 and 1033 share row 9: two distinct rows saw bucket 3)
    rec = np.frombuffer(notify.tobytes(), dtype=np.dtype([("glob_id", "<u8"), ("nqrys_5s", "<u4"), ("total_resp_5sec", "<u4"), ("nconns", "<u4"),
                                                        ("nconns_active", "<u4"), ("ntasks", "<u4"), ("p95_5s", "<u4"), ("p95_5m", "<u4")]), count=1)[0]
    assert int(rec["glob_id"]) == 0xabcdef0123 and int(rec["nqrys_5s"]) == nq and int(rec["total_resp_5sec"]) == int(last[:, 1].sum()) & 0xFFFFFFFF
    assert int(rec["nconns_active"]) == 3 and int(rec["p95_5s"]) == out.p95_ms[0] and int(rec["p95_5m"]) == out.p95_ms[1]
    assert notify[79] == 2
