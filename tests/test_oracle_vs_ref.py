"""Randomised cross-check of the plain-C oracle against the reference's own compiled headers (oracle/_ref).  Skipped where
oracle/_ref is absent (it is built only where /root/reference exists; the committed golden fixtures cover the rest)."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("kind", range(10))
def test_bucket_and_hist_random(oracle, reflib, kind):
    L, R = oracle.lib(), reflib
    rng = np.random.default_rng(100 + kind)
    h = R.ref_hist_new(kind)
    nb = R.ref_hist_nbuckets(h)
    assert nb == L.gyo_hist_nbuckets(kind)
    vals = np.concatenate([
        rng.integers(-50, 200, 3000), rng.integers(-10, 6_000_000, 3000), (rng.lognormal(3, 2.5, 3000)).astype(np.int64),
        np.array([2**31 - 1, 2**31, 2**31 + 5, -2**31, 2**40 + 17, -2**40, 2**62], dtype=np.int64)]).astype(np.int64)
    if kind == 8:
        vals = rng.integers(-128, 128, 3000).astype(np.int64)
    bo = np.zeros(len(vals), dtype=np.uint32)
    br = np.zeros(len(vals), dtype=np.uint32)
    L.gyo_bucket_many(kind, oracle.ptr(vals, oracle.i64p), len(vals), oracle.ptr(bo, oracle.u32p))
    R.ref_hist_bucket_of_many(h, oracle.ptr(vals, oracle.i64p), len(vals), oracle.ptr(br, oracle.u32p))
    assert (bo == br).all()
    for i in range(nb + 2):
        assert L.gyo_bucket_max_threshold(kind, i) == R.ref_hist_bucket_max_threshold(h, i)

    oh = oracle.Hist()
    L.gyo_hist_init(C.byref(oh), kind)
    h2 = R.ref_hist_new(kind)
    oh2 = oracle.Hist()
    L.gyo_hist_init(C.byref(oh2), kind)
    half = len(vals) // 2
    a, b = np.ascontiguousarray(vals[:half]), np.ascontiguousarray(vals[half:])
    R.ref_hist_add_many(h, oracle.ptr(a, oracle.i64p), len(a))
    R.ref_hist_add_many(h2, oracle.ptr(b, oracle.i64p), len(b))
    L.gyo_hist_add_many(C.byref(oh), oracle.ptr(a, oracle.i64p), len(a))
    L.gyo_hist_add_many(C.byref(oh2), oracle.ptr(b, oracle.i64p), len(b))
    R.ref_hist_merge(h, h2)           # add_histogram
    L.gyo_hist_merge(C.byref(oh), C.byref(oh2))

    counts = np.zeros(nb, dtype=np.uint64)
    sums = np.zeros(nb, dtype=np.int64)
    total, maxv = C.c_uint64(), C.c_int64()
    R.ref_hist_serialized(h, oracle.ptr(counts, oracle.u64p), oracle.ptr(sums, oracle.i64p), C.byref(total), C.byref(maxv))
    assert [oh.stats[i].count for i in range(nb)] == counts.tolist()
    assert [oh.stats[i].sum for i in range(nb)] == sums.tolist()
    assert oh.total_count == total.value and oh.max_val_seen == maxv.value

    pcts = np.array([0.001, 1, 10, 25, 50, 75, 90, 95, 99, 99.9, 99.999, 100], dtype=np.float32)
    pv = np.zeros(len(pcts), dtype=np.int64)
    ps = np.zeros(len(pcts), dtype=np.int64)
    pc = np.zeros(len(pcts), dtype=np.uint64)
    avg = C.c_float()
    R.ref_hist_percentiles(h, oracle.ptr(pcts, oracle.f32p), len(pcts), oracle.ptr(pv, oracle.i64p), oracle.ptr(ps, oracle.i64p),
                           oracle.ptr(pc, oracle.u64p), C.byref(total), C.byref(maxv), C.byref(avg))
    pd = (oracle.HistData * len(pcts))()
    for i, p in enumerate(pcts):
        pd[i].percentile = float(p)
    oavg = C.c_float()
    L.gyo_hist_percentiles(C.byref(oh), pd, len(pcts), None, None, C.byref(oavg))
    assert [d.data_value for d in pd] == pv.tolist()
    assert [d.sum for d in pd] == ps.tolist()
    assert [d.count for d in pd] == pc.tolist()
    assert np.float32(oavg.value).tobytes() == np.float32(avg.value).tobytes()
    R.ref_hist_free(h)
    R.ref_hist_free(h2)


def test_percentile_float_cutoff_large_counts(oracle, reflib):
    """ncutoff = size_t * float: exercise counts above 2^24 where the float product rounds (gy_statistics.h:757-758)."""
    L, R = oracle.lib(), reflib
    h = R.ref_hist_new(0)
    oh = oracle.Hist()
    L.gyo_hist_init(C.byref(oh), 0)
    rng = np.random.default_rng(7)
    vals = rng.integers(0, 1200, 40_000_000 // 64).astype(np.int64)
    for _ in range(64):  # 40M adds -> total_count > 2^25
        R.ref_hist_add_many(h, oracle.ptr(vals, oracle.i64p), len(vals))
        L.gyo_hist_add_many(C.byref(oh), oracle.ptr(vals, oracle.i64p), len(vals))
    pcts = np.array([33.333, 50, 66.6667, 99.9999, 100], dtype=np.float32)
    pv = np.zeros(len(pcts), dtype=np.int64)
    ps = np.zeros(len(pcts), dtype=np.int64)
    pc = np.zeros(len(pcts), dtype=np.uint64)
    total, maxv, avg = C.c_uint64(), C.c_int64(), C.c_float()
    R.ref_hist_percentiles(h, oracle.ptr(pcts, oracle.f32p), len(pcts), oracle.ptr(pv, oracle.i64p), oracle.ptr(ps, oracle.i64p),
                           oracle.ptr(pc, oracle.u64p), C.byref(total), C.byref(maxv), C.byref(avg))
    pd = (oracle.HistData * len(pcts))()
    for i, p in enumerate(pcts):
        pd[i].percentile = float(p)
    oavg = C.c_float()
    L.gyo_hist_percentiles(C.byref(oh), pd, len(pcts), None, None, C.byref(oavg))
    assert [d.data_value for d in pd] == pv.tolist() and [d.count for d in pd] == pc.tolist()
    assert np.float32(oavg.value).tobytes() == np.float32(avg.value).tobytes()
    R.ref_hist_free(h)


def test_hashes_random(oracle, reflib):
    L, R = oracle.lib(), reflib
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(1, 13))
        w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        seed = int(rng.integers(0, 2**32))
        assert L.gyo_jhash2(oracle.ptr(w, oracle.u32p), n, seed) == R.ref_jhash2(oracle.ptr(w, oracle.u32p), n, seed)
        b = w.tobytes()[: int(rng.integers(0, 4 * n + 1))]
        assert L.gyo_jhash(b, len(b), seed) == R.ref_jhash(b, len(b), seed)
        k = int(rng.integers(0, 2**63))
        assert L.gyo_get_uint64_hash(k) == R.ref_get_uint64_hash(k)
        assert L.gyo_get_uint32_hash(k & 0xFFFFFFFF) == R.ref_get_uint32_hash(k & 0xFFFFFFFF)
        a, c = int(rng.integers(0, 2**63)), int(rng.integers(0, 2**63))
        assert L.gyo_machine_id_hash(a, c) == R.ref_machine_id_hash(a, c)
        c6, s6 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        cip = bytes(rng.integers(0, 256, 16, dtype=np.uint8).tolist()) if c6 else int(rng.integers(0, 2**32))
        sip = bytes(rng.integers(0, 256, 16, dtype=np.uint8).tolist()) if s6 else int(rng.integers(0, 2**32))
        cb, cf = oracle.ip_bytes(cip)
        sb, sf = oracle.ip_bytes(sip)
        cp, sp = int(rng.integers(0, 65536)), int(rng.integers(0, 65536))
        ino = int(rng.integers(0, 2**40))
        for ign in (0, 1):
            assert L.gyo_ip_port_hash(cb, cf, cp, ign) == R.ref_ip_port_hash(cb, cf, cp, ign)
            assert L.gyo_ns_ip_port_hash(sb, sf, sp, ino, ign) == R.ref_ns_ip_port_hash(sb, sf, sp, ino, ign)
        assert L.gyo_pair_ip_port_hash(cb, cf, cp, sb, sf, sp) == R.ref_pair_ip_port_hash(cb, cf, cp, sb, sf, sp)


def test_topn_random(oracle, reflib):
    L, R = oracle.lib(), reflib
    rng = np.random.default_rng(11)
    for n, m in [(10, 1000), (50, 3000), (10, 7), (1, 100)]:
        v = rng.integers(0, 500, m, dtype=np.uint64)
        o1 = np.zeros(n, dtype=np.uint64)
        o2 = np.zeros(n, dtype=np.uint64)
        k1 = L.gyo_topn_u64(oracle.ptr(v, oracle.u64p), m, n, oracle.ptr(o1, oracle.u64p))
        k2 = R.ref_topn_u64(oracle.ptr(v, oracle.u64p), m, n, oracle.ptr(o2, oracle.u64p))
        assert k1 == k2 and o1[:k1].tolist() == o2[:k2].tolist()


def test_struct_sizes(reflib):
    assert [reflib.ref_sizeof(i) for i in range(7)] == [24, 32, 64, 40, 16, 280, 32]


def test_keyed_resp_loop_of_the_reference_classes_matches_the_port(oracle):
    """bench.py's cpu_baseline kind "reference": the reference's own GY_HISTOGRAM<int64_t, RESP_TIME_HASH> behind an unordered_map with
    GY_JHASHER (oracle/ref_glue.cc) must count exactly what the plain-C port counts on the same event bytes"""
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_keyed_new"):
        pytest.skip("oracle/_ref not built (no /root/reference)")
    from gyeeta_amd import wire
    from tests import helpers
    rng = np.random.default_rng(1)
    k = R.ref_keyed_new()
    orc = oracle.OracleEngine(64, enable_td=False)
    for h in range(3):
        s = np.arange(6)
        ns, pt, g = wire.listener_netns(h, s), wire.listener_port(s), wire.glob_id(np.full(6, h), s)
        for i in range(6):
            assert R.ref_keyed_register(k, h, int(ns[i]), int(pt[i])) == h * 6 + i
            orc.register(h, int(g[i]), int(ns[i]), int(pt[i]))
    added = 0
    for h in range(3):
        ev = helpers.make_resp_events(rng, h, 4000, 6)
        b = ev.tobytes()
        buf = np.frombuffer(b, dtype=np.uint8)
        sh, sf = np.array([h], dtype=np.uint32), np.array([0], dtype=np.uint64)
        added += R.ref_keyed_resp_batch(k, buf.ctypes.data, 4000, oracle.ptr(sh, oracle.u32p), oracle.ptr(sf, oracle.u64p), 1)
        orc.resp_batch(b, [h], [0])
    assert added == orc.counters()["accepted"]
    assert [R.ref_keyed_total(k, i) for i in range(18)] == orc.hist()[:18, 15, 0].tolist()
    # the multi-host batch form, single- and multi-threaded (hosts cut into ranges), must count the same again
    parts = [helpers.make_resp_events(rng, h, 2500, 6) for h in range(3)]
    raw = b"".join(p.tobytes() for p in parts)
    buf = np.frombuffer(raw, dtype=np.uint8)
    sh, sf = np.arange(3, dtype=np.uint32), (np.arange(3) * 2500).astype(np.uint64)
    a1 = R.ref_keyed_resp_batch(k, buf.ctypes.data, 7500, oracle.ptr(sh, oracle.u32p), oracle.ptr(sf, oracle.u64p), 3)
    orc.resp_batch(raw, [0, 1, 2], [0, 2500, 5000])
    assert added + a1 == orc.counters()["accepted"] and a1 > 7000
    if hasattr(R, "ref_keyed_resp_batch_mt"):
        a2 = R.ref_keyed_resp_batch_mt(k, buf.ctypes.data, 7500, oracle.ptr(sh, oracle.u32p), oracle.ptr(sf, oracle.u64p), 3, 3)
        orc.resp_batch(raw, [0, 1, 2], [0, 2500, 5000])
        assert a2 == a1
        assert [R.ref_keyed_total(k, i) for i in range(18)] == orc.hist()[:18, 15, 0].tolist()
    R.ref_keyed_free(k)


def test_slab_percentile_rule_vs_reference_container(reflib, oracle):
    """TIME_HISTOGRAM::get_stats ranks with folly::detail::SlabHistogramBuckets::getPercentileBucketIdx -- the reference's OWN in-tree
    copy (thirdparty/SlabHistogramBucket.h:165-240), compiled here and constructed as TIME_HISTOGRAM constructs it
    (common/gy_statistics.h:1106-1108).  The oracle's (and the engine's) percentile rule and bucket numbering for the time levels are
    pinned against it; what stays unpinned of the multi-level windows is folly's ring arithmetic alone."""
    R, L = reflib, oracle.lib()
    if not hasattr(R, "ref_slab_percentile_idx"):
        pytest.skip("oracle/_ref built without the slab container")
    assert R.ref_slab_num_buckets() == 15 == L.gyo_hist_nbuckets(oracle.RESP_TIME_HASH)
    rng = np.random.default_rng(2024)
    vals = np.concatenate([np.arange(-3, 40), [59, 60, 61, 999, 1000, 1001, 14999, 15000, 15001, 10**6, 2**40, -2**40],
                           rng.integers(-10, 20000, 2000)])
    for v in vals:
        assert R.ref_slab_bucket_idx(int(v)) == L.gyo_bucket(oracle.RESP_TIME_HASH, int(v)), v
    pcts = [0.0, 1e-9, 0.001, 0.25, 0.5, 0.75, 0.95, 0.99, 0.999, 0.9999, 1.0]
    for trial in range(400):
        kind = trial % 4
        if kind == 0:
            counts = rng.integers(0, 1000, 15)
        elif kind == 1:
            counts = rng.integers(0, 3, 15) * rng.integers(0, 10**6, 15)      # many empty buckets
        elif kind == 2:
            counts = np.zeros(15, dtype=np.int64)
            counts[rng.integers(0, 15)] = rng.integers(1, 10**9)              # a single bucket
        else:
            counts = rng.integers(0, 2**40, 15)                               # counts beyond 2^32: double rounding of the fractions
        counts = counts.astype(np.uint64)
        for pct in pcts + rng.random(5).tolist():
            want = R.ref_slab_percentile_idx(oracle.ptr(counts, oracle.u64p), 15, float(pct))
            assert L.gyo_slab_percentile_idx(oracle.ptr(counts, oracle.u64p), 15, float(pct)) == want, (counts.tolist(), pct)
    empty = np.zeros(15, dtype=np.uint64)
    assert R.ref_slab_percentile_idx(oracle.ptr(empty, oracle.u64p), 15, 0.95) == 1 == L.gyo_slab_percentile_idx(oracle.ptr(empty, oracle.u64p), 15, 0.95)


def test_cluster_state_one_vs_reference(reflib, oracle):
    """comm::MS_CLUSTER_STATE::STATE_ONE (common/gy_comm_proto.h:3183-3213): counter order and add_stats == gyo_cluster_state_one /
    gyo_cluster_state_add (what the engine's all-reduce of the cluster rows computes)"""
    R, L = reflib, oracle.lib()
    if not hasattr(R, "ref_state_one_add"):
        pytest.skip("oracle/_ref built without gy_comm_proto")
    assert R.ref_state_one_sizeof() == 48  # 11 x u32, alignas(8)
    names = [n for n, _ in oracle.ClusterStateOne._fields_]
    assert names == ["nhosts", "ntasks_issue", "ntaskissue_hosts", "ntasks", "nsvc_issue", "nsvcissue_hosts", "nsvc", "total_qps", "svc_net_mb",
                     "ncpu_issue", "nmem_issue"]
    probe = np.arange(100, 111, dtype=np.uint32)
    out = np.zeros(11, dtype=np.uint32)
    R.ref_state_one_fields(oracle.ptr(probe, oracle.u32p), oracle.ptr(out, oracle.u32p))
    assert out.tolist() == probe.tolist()  # members in declaration order, no holes
    rng = np.random.default_rng(8)
    for _ in range(50):
        a = rng.integers(0, 2**32, 11, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2**32, 11, dtype=np.uint64).astype(np.uint32)
        ra = a.copy()
        R.ref_state_one_add(oracle.ptr(ra, oracle.u32p), oracle.ptr(b, oracle.u32p))
        oa, ob = oracle.ClusterStateOne(*a.tolist()), oracle.ClusterStateOne(*b.tolist())
        L.gyo_cluster_state_add(C.byref(oa), C.byref(ob))
        assert list(oa.as_tuple()) == ra.tolist()  # wraps modulo 2^32 like the reference's uint32_t sums


def test_listen_summ_stats_and_cluster_state_one_equal_the_reference_classes(oracle, reflib):
    """LISTEN_SUMM_STATS<int>::update (server/gy_msocket.h:856-868) and CLUSTER_STATE_ONE::update_from_state
    (server/gy_mconnhdlr.cc:16034-16049): the reference's own classes -- their text cut out of files that cannot be compiled here and
    built into oracle/_ref by oracle/build_ref.sh -- against the oracle's restatements (gyo_listener_state_rollup,
    gyo_cluster_state_update) on random record batches incl. invalid states, zero / huge query counts and counter wrap-around"""
    import ctypes as C
    import numpy as np
    from gyeeta_amd import wire
    if not hasattr(reflib, "ref_has_summ_stats") or not reflib.ref_has_summ_stats():
        import pytest
        pytest.skip("oracle/_ref built without the two server-side classes")
    L = oracle.lib()
    rng = np.random.default_rng(1234)
    for trial in range(40):
        n = int(rng.integers(0, 600))
        rec = np.zeros(n, dtype=wire.LISTENER_STATE_NOTIFY)
        rec["glob_id"] = rng.integers(1, 1 << 62, n, dtype=np.uint64)
        big = trial % 5 == 0
        rec["nqrys_5s"] = rng.integers(0, (1 << 32) if big else 5000, n, dtype=np.uint64).astype(np.uint32)
        rec["nqrys_5s"][rng.random(n) < 0.2] = 0
        for f in ("nconns_active", "curr_kbytes_inbound", "curr_kbytes_outbound", "ser_errors"):
            rec[f] = rng.integers(0, (1 << 31) if big else 100000, n, dtype=np.uint64).astype(np.uint32)
        rec["curr_state"] = rng.integers(0, 9 if trial % 3 == 0 else 6, n)  # states above STATE_DOWN are skipped by the caller
        raw = rec.tobytes()
        buf = np.frombuffer(raw, dtype=np.uint8) if n else np.zeros(1, dtype=np.uint8)
        want = (C.c_int32 * 13)()
        reflib.ref_listen_summ_update(buf.ctypes.data, n, want)
        summ = oracle.ListenSummStats()
        nerr = C.c_int(0)
        L.gyo_listener_state_rollup(oracle.ptr(buf, oracle.u8p), n, C.cast(buf.ctypes.data + n * 88, oracle.u8p), C.byref(summ), C.byref(nerr))
        assert list(summ.as_tuple()) == list(want), (trial, summ.as_tuple(), list(want))
        assert nerr.value == int((rec["curr_state"] > 5).sum())
        # one host into a cluster row that already holds something
        st0 = rng.integers(0, 1 << 31, 11, dtype=np.uint64).astype(np.uint32)
        a = st0.copy()
        hs = [int(x) for x in rng.integers(0, 5000, 4)] + [int(rng.integers(0, 2)), int(rng.integers(0, 2))]
        if trial % 4 == 0:
            hs[0] = 0
            hs[2] = 0
        reflib.ref_cluster_state_update(oracle.ptr(a, oracle.u32p), hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], want)
        b = oracle.ClusterStateOne()
        for i, (name, _) in enumerate(b._fields_):
            setattr(b, name, int(st0[i]))
        L.gyo_cluster_state_update(C.byref(b), hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], C.byref(summ))
        assert [getattr(b, nm) for nm, _ in b._fields_] == a.tolist(), (trial, hs)
