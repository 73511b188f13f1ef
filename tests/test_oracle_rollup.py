"""The roll-up oracle (oracle/gy_oracle_rollup.c: 64-bit counters) against the per-service digest oracle it extends (oracle/gy_oracle.c,
32-bit counters): with weights that fit both, the two must agree cluster for cluster -- merging values, merging a digest's clusters, and
quantiles -- so that the GPU roll-up's bit-exactness against the 64-bit form is also bit-exactness against the pinned 32-bit definition.
Plus the properties a roll-up must have: totals add up, min / max cover the members, fold order matters only through the definition."""
import ctypes as C

import numpy as np
import pytest


def _vals(rng, n, mu=3.0):
    return np.ascontiguousarray(np.minimum(np.floor(rng.lognormal(mu, 1.5, n)), 1e6).astype(np.int32))


def _td_equal(L, d32, d64):
    return list(d32.sum) == list(d64.sum) and [int(c) for c in d32.cnt] == [int(c) for c in d64.cnt]


def test_td64_matches_td32_on_values_and_digest_merges(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for trial in range(20):
        a32, a64 = oracle.TDigest(), oracle.TD64()
        L.gyo_td_init(C.byref(a32))
        L.gyo_td64_init(C.byref(a64))
        for _ in range(int(rng.integers(1, 6))):  # several value merges in a row
            v = _vals(rng, int(rng.integers(1, 3000)), mu=float(rng.normal(3, 1)))
            L.gyo_td_merge_values(C.byref(a32), oracle.ptr(v, oracle.i32p), len(v))
            L.gyo_td64_merge_values(C.byref(a64), oracle.ptr(v, oracle.i32p), len(v))
            assert _td_equal(L, a32, a64) and a32.vmin == a64.vmin and a32.vmax == a64.vmax
        # another digest's clusters as weighted points: gyo_td_merge_digest vs the service form with an empty buffer
        b = oracle.TDBuffered()
        L.gyo_tdb_init(C.byref(b))
        v = _vals(rng, 2500, mu=float(rng.normal(3, 1)))
        L.gyo_td_merge_values(C.byref(b.d), oracle.ptr(v, oracle.i32p), len(v))
        L.gyo_td_merge_digest(C.byref(a32), C.byref(b.d))
        L.gyo_td64_merge_service(C.byref(a64), C.byref(b))
        assert _td_equal(L, a32, a64) and a32.vmin == a64.vmin and a32.vmax == a64.vmax
        for q in (0.0, 0.01, 0.5, 0.95, 0.999, 1.0):
            assert L.gyo_td_quantile(C.byref(a32), q) == L.gyo_td64_quantile(C.byref(a64), q)


def test_td64_service_merge_is_clusters_then_buffer(oracle):
    """a service contributes its clusters first and its buffered values second: equal to doing the two steps by hand (the buffer is
    merged into the GROUP digest, not into the member's own)"""
    L = oracle.lib()
    rng = np.random.default_rng(6)
    b = oracle.TDBuffered()
    L.gyo_tdb_init(C.byref(b))
    for n in (500, 500, 300):  # 500 buffered, 1000 > 896: one merge, then 300 buffered
        v = _vals(rng, n)
        L.gyo_tdb_add_batch(C.byref(b), oracle.ptr(v, oracle.i32p), len(v))
    assert b.npend == 300 and L.gyo_td_total(C.byref(b.d)) == 1000
    seed = _vals(rng, 4000, mu=2.0)
    g1, g2 = oracle.TD64(), oracle.TD64()
    for g in (g1, g2):
        L.gyo_td64_init(C.byref(g))
        L.gyo_td64_merge_values(C.byref(g), oracle.ptr(seed, oracle.i32p), len(seed))
    L.gyo_td64_merge_service(C.byref(g1), C.byref(b))
    clusters_only = oracle.TDBuffered()
    L.gyo_tdb_init(C.byref(clusters_only))
    clusters_only.d = b.d
    L.gyo_td64_merge_service(C.byref(g2), C.byref(clusters_only))
    pend = np.ascontiguousarray(np.array(b.pend[:b.npend], dtype=np.int32))
    L.gyo_td64_merge_values(C.byref(g2), oracle.ptr(pend, oracle.i32p), len(pend))
    assert list(g1.sum) == list(g2.sum) and list(g1.cnt) == list(g2.cnt) and g1.vmin == g2.vmin and g1.vmax == g2.vmax
    assert L.gyo_td64_total(C.byref(g1)) == 4000 + L.gyo_tdb_total(C.byref(b))


def test_td64_rollup_totals_minmax_and_rank_error(oracle):
    """a group of 300 services with different scales: the roll-up's weight is the sum of the members', its extremes cover theirs, and
    its quantiles rank within 1 % of the pooled exact sort (the group digest is as good as a service's)"""
    L = oracle.lib()
    rng = np.random.default_rng(7)
    g = oracle.TD64()
    L.gyo_td64_init(C.byref(g))
    pooled = []
    for s in range(300):
        b = oracle.TDBuffered()
        L.gyo_tdb_init(C.byref(b))
        mu = float(rng.normal(3, 1))
        for _ in range(int(rng.integers(1, 5))):
            v = _vals(rng, int(rng.integers(50, 900)), mu=mu)
            pooled.append(v)
            L.gyo_tdb_add_batch(C.byref(b), oracle.ptr(v, oracle.i32p), len(v))
        L.gyo_td64_merge_service(C.byref(g), C.byref(b))
    x = np.sort(np.concatenate(pooled))
    assert L.gyo_td64_total(C.byref(g)) == len(x) and g.vmin == int(x[0]) and g.vmax == int(x[-1])
    for q in (0.05, 0.25, 0.5, 0.9, 0.99):
        v = L.gyo_td64_quantile(C.byref(g), q)
        lo, hi = np.searchsorted(x, v, side="left") / len(x), np.searchsorted(x, v, side="right") / len(x)
        err = 0.0 if lo <= q <= hi else min(abs(lo - q), abs(hi - q))
        assert err <= 0.01, (q, v, err)


def test_active_conn_and_pair_oracles_against_numpy(oracle):
    from gyeeta_amd import wire
    L = oracle.lib()
    rng = np.random.default_rng(8)
    rec = wire.synth_active_conns(rng, 3000, 2, 9)
    p32 = np.zeros((4, 65536), dtype=np.uint32)
    p64 = np.zeros((4, 65536), dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    buf = np.frombuffer(rec.tobytes(), dtype=np.uint8)
    L.gyo_active_conn_sketch_batch(oracle.ptr(buf, oracle.u8p), len(rec), oracle.ptr(p32, oracle.u32p), oracle.ptr(p64, oracle.u64p), oracle.ptr(out, oracle.u64p))
    local = (rec["flags"] & wire.ACTIVE_FLAG_REMOTE_LISTEN) == 0
    assert out.tolist() == [int(local.sum()), int((~local).sum())]
    assert p32.sum(axis=1).tolist() == [int(rec["active_conns"][local].sum())] * 4
    assert p64.sum(axis=1).tolist() == [int(rec["bytes_sent"][local].sum() + rec["bytes_received"][local].sum())] * 4
    # Count-Min never under-estimates: a pair's estimate (min over the rows) >= its exact total
    g, t = int(rec["listener_glob_id"][local][0]), int(rec["cli_aggr_task_id"][local][0])
    w = np.array([g & 0xFFFFFFFF, g >> 32, t & 0xFFFFFFFF, t >> 32], dtype=np.uint32)
    est = L.gyo_cms_query(oracle.ptr(p32, oracle.u32p), oracle.ptr(w, oracle.u32p), 4)
    sel = local & (rec["listener_glob_id"] == g) & (rec["cli_aggr_task_id"] == t)
    assert est >= int(rec["active_conns"][sel].sum())
