"""The roll-up oracle (oracle/gy_oracle_rollup.c: 64-bit counters) against the per-service digest oracle it extends (oracle/gy_oracle.c,
32-bit counters): with weights that fit both, the two must agree cluster for cluster -- merging values, merging a digest's clusters, and
quantiles -- so that the GPU roll-up's bit-exactness against the 64-bit form is also bit-exactness against the pinned 32-bit definition.
Plus the properties a roll-up must have: totals add up, min / max cover the members.  Round 6: the roll-up itself is the union by value
bin (gyo_tdbins_*): its finish against a point-by-point restatement, conservation, independence of the members' order, rank error."""
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


# ---------------------------------------------------------------- round 6: the roll-up is the union by value bin (gyo_tdbins_*)
def _rank_err(x, v, q):
    lo, hi = np.searchsorted(x, v, side="left") / len(x), np.searchsorted(x, v, side="right") / len(x)
    return 0.0 if lo <= q <= hi else min(abs(lo - q), abs(hi - q))


def test_value_bins_cover_the_domain_in_order(oracle):
    L = oracle.lib()
    v = np.arange(0, 1 << 20, dtype=np.uint32)
    b = np.array([L.gyo_td_value_bin(int(x)) for x in v[:: 7]])
    assert (np.diff(b) >= 0).all() and b[0] == 0
    assert [L.gyo_td_value_bin(x) for x in (0, 1, 1023, 1024, 1039, 1040, 2047, 2048)] == [0, 1, 1023, 1024, 1024, 1025, 1087, 1088]
    assert L.gyo_td_value_bin((1 << 26) - 1) == oracle.TD_BINS - 1 == L.gyo_td_value_bin(0xFFFFFFFF)
    # a cell is at most 1 / 64 of its lower edge wide
    for x in (1024, 5000, 99999, 1 << 19):
        k = L.gyo_td_value_bin(x)
        same = [y for y in range(x, x + x // 32) if L.gyo_td_value_bin(y) == k]
        assert len(same) <= x // 64 + 1


def test_tdbins_finish_equals_the_point_by_point_restatement(oracle):
    """gyo_tdbins_finish against the definition spelled out point by point in Python: every unit mid-point of every bin gets its cluster
    from gyo_td_cluster, a bin's sum is shared by floor(sum r1 / w) - floor(sum r0 / w) over the runs of equal cluster (Python integers)"""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    for case in range(6):
        b = oracle.TDBins()
        L.gyo_tdbins_init(C.byref(b))
        if case < 4:
            v = _vals(rng, int(rng.integers(1, 4000)), mu=float(rng.uniform(0.5, 8.0)))
            L.gyo_tdbins_add_values(C.byref(b), oracle.ptr(v, oracle.i32p), len(v))
        else:  # few heavy bins with sums that are not multiples of the weight: every cluster boundary cuts a bin
            for k in rng.integers(0, oracle.TD_BINS, 5):
                b.cnt[int(k)] = int(rng.integers(500, 3000))
                b.sum[int(k)] = int(b.cnt[int(k)] * int(k) - rng.integers(0, 400))
        out = oracle.TD64()
        L.gyo_tdbins_finish(C.byref(b), C.byref(out))
        N = sum(b.cnt)
        want_s, want_c, W = [0] * oracle.TD_NB, [0] * oracle.TD_NB, 0
        for k in range(oracle.TD_BINS):
            w, sm = int(b.cnt[k]), int(b.sum[k])
            if not w:
                continue
            cl = [L.gyo_td_cluster(2 * (W + r) + 1, 2 * N) for r in range(w)]
            r0 = 0
            for r in range(1, w + 1):
                if r == w or cl[r] != cl[r0]:
                    want_s[cl[r0]] += sm * r // w - sm * r0 // w
                    want_c[cl[r0]] += r - r0
                    r0 = r
            W += w
        assert list(out.cnt) == want_c and list(out.sum) == want_s, case
        assert sum(out.sum) == sum(b.sum) and sum(out.cnt) == N


@pytest.mark.parametrize("name,nsvc,mus,sig", [("mixed", 300, (3, 1), 0.8), ("tight", 300, (1.5, 0.1), 0.3), ("seconds", 200, (7.5, 0.5), 1.0),
                                               ("one-heavy-value", 300, (0.7, 0.05), 0.2)])
def test_tdbins_rollup_totals_order_and_rank_error(oracle, name, nsvc, mus, sig):
    """a group of services rolled up by value bin: weight and sum of the members are conserved exactly, the extremes cover theirs, the
    order of the members does not matter, and the quantiles rank within 1 % of the pooled exact sort (tolerance of the per-service
    digests; here: a few 10^-4) -- on one level and on two (services -> 10 host slabs -> one)"""
    L = oracle.lib()
    rng = np.random.default_rng(7)
    svcs, pooled = [], []
    for s in range(nsvc):
        b = oracle.TDBuffered()
        L.gyo_tdb_init(C.byref(b))
        mu = float(rng.normal(*mus))
        for _ in range(int(rng.integers(1, 6))):
            v = np.ascontiguousarray(np.clip(rng.lognormal(mu, sig, int(rng.integers(50, 900))), 0, 1e6).astype(np.int32))
            pooled.append(v)
            L.gyo_tdb_add_batch(C.byref(b), oracle.ptr(v, oracle.i32p), len(v))
        svcs.append(b)
    x = np.sort(np.concatenate(pooled))
    one = oracle.rollup_services(svcs)
    assert L.gyo_td64_total(C.byref(one)) == len(x) and sum(one.sum) == int(x.sum()) and one.vmin == int(x[0]) and one.vmax == int(x[-1])
    rev = oracle.rollup_services(svcs[::-1])
    assert list(rev.sum) == list(one.sum) and list(rev.cnt) == list(one.cnt)
    hosts = [oracle.rollup_services(svcs[h::10]) for h in range(10)]
    two = oracle.rollup_slabs(hosts)
    assert L.gyo_td64_total(C.byref(two)) == len(x) and sum(two.sum) == int(x.sum()) and two.vmin == int(x[0]) and two.vmax == int(x[-1])
    # the clusters' means rise with the bins; the pieces of ONE bin that is cut by cluster boundaries carry integer shares of its sum, so their
    # means differ by less than one unit per point of the piece among themselves (all of them lie inside the bin)
    means = np.array([s_ / c_ for s_, c_ in zip(one.sum, one.cnt) if c_])
    assert (np.diff(means) > -np.maximum(1.0, means[1:] / 64)).all()
    assert np.abs(np.diff(means)[np.diff(means) < 0]).sum() < 1.0 + means[-1] / 64
    for q in (0.01, 0.05, 0.25, 0.5, 0.9, 0.95, 0.99, 0.999):
        for d in (one, two):
            v = L.gyo_td64_quantile(C.byref(d), q)
            assert _rank_err(x, v, q) <= 0.01, (name, q, v)


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
