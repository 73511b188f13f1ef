"""Oracle of the multi-level response histogram (oracle/gy_oracle_levels.c = folly BucketedTimeSeries / MultiLevelTimeSeries as
TIME_HISTOGRAM drives them, common/gy_statistics.h:1082-1551).  folly is not in /root/reference and the reference's own test at
this boundary only prints, so parity is UNPINNED; what can be checked is the restatement against the definition of the ring
("a level holds the adds from the start of its oldest live bucket up to now"), written here a second, independent way, and the
in-tree percentile rule (thirdparty/SlabHistogramBucket.h:165-240)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

LEVEL_SECS = [5, 300, 5 * 24 * 3600, 0]
NB = 10


def window_start(t, dur, nb):
    """first second still covered by a ring of nb buckets over dur seconds after update(t)"""
    nb = min(nb, dur)
    c = (t % dur) * nb // dur
    j = (c + 1) % nb
    start = (t // dur) * dur + -(-(j * dur) // nb)  # ceil(j*dur/nb), folly getBucketInfo
    return start if start <= t else start - dur


def brute_level(adds, tq, level):
    """adds: list of (t, sum, count) with t <= tq, monotonic"""
    dur = LEVEL_SECS[level]
    lo = window_start(tq, dur, NB) if dur else -1
    return sum(s for t, s, c in adds if t >= lo), sum(c for t, s, c in adds if t >= lo)


def test_level_seconds():
    L = oracle.lib()
    assert [L.gyo_mlh_level_seconds(i) for i in range(4)] == LEVEL_SECS


@pytest.mark.parametrize("cadence", ["regular5", "jitter", "gaps"])
def test_bucketed_series_matches_definition(cadence):
    L = oracle.lib()
    rng = np.random.default_rng(11)
    series = [oracle.BTS() for _ in range(4)]
    for lv, s in enumerate(series):
        L.gyo_bts_init(C.byref(s), NB, LEVEL_SECS[lv])
    assert [s.nbuckets for s in series] == [5, 10, 10, 0]  # nBuckets is rounded down to the duration (5 s level), all-time has none
    t = 1_700_000_003
    adds = []
    for step in range(600):
        if cadence == "regular5":
            t += 5
        elif cadence == "jitter":
            t += int(rng.integers(1, 9))
        else:
            t += int(rng.choice([5, 5, 5, 40, 301, 3600, 43200 * 3, 5 * 24 * 3600 + 7]))
        if rng.random() < 0.8:
            sm, cnt = int(rng.integers(0, 10**6)), int(rng.integers(1, 1000))
            adds.append((t, sm, cnt))
            for s in series:
                assert L.gyo_bts_add(C.byref(s), t, sm, cnt) == 1
        tq = t + int(rng.integers(0, 3)) if step % 7 == 0 else t
        for lv, s in enumerate(series):
            L.gyo_bts_update(C.byref(s), tq)
            assert (s.tot_sum, s.tot_cnt) == brute_level(adds, tq, lv), (cadence, step, lv)
            if LEVEL_SECS[lv]:
                assert sum(s.bsum) == s.tot_sum and sum(s.bcnt) == s.tot_cnt
        t = max(t, tq)


def test_regular_cadence_level0_is_last_window():
    L = oracle.lib()
    h = oracle.MLHist()
    L.gyo_mlh_init(C.byref(h), oracle.RESP_TIME_HASH, NB)
    rng = np.random.default_rng(5)
    t = 1_700_000_000
    cum = np.zeros((16, 2), dtype=np.int64)
    out = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
    for w in range(80):
        t += 5
        stats = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
        if w % 9 != 4:  # some idle windows
            stats["count"][:15] = rng.integers(0, 50, 15)
            stats["sum"][:15] = stats["count"][:15] * rng.integers(1, 2000, 15)
        L.gyo_mlh_add_hist(C.byref(h), t, stats.ctypes.data, 1)
        L.gyo_mlh_flush(C.byref(h), t)
        cum[:, 0] += stats["count"].astype(np.int64)
        cum[:, 1] += stats["sum"]
        L.gyo_mlh_level(C.byref(h), 0, out.ctypes.data)
        assert out["count"].tolist() == stats["count"].tolist() and out["sum"].tolist() == stats["sum"].tolist()
        L.gyo_mlh_level(C.byref(h), 3, out.ctypes.data)
        assert out["count"].astype(np.int64).tolist() == cum[:, 0].tolist() and out["sum"].tolist() == cum[:, 1].tolist()
    L.gyo_mlh_flush(C.byref(h), t + 5)  # nothing arrives for one window: "last 5 seconds" is empty again
    L.gyo_mlh_level(C.byref(h), 0, out.ctypes.data)
    assert int(out["count"].sum()) == 0


def test_slab_percentile_rule():
    L = oracle.lib()
    counts = np.array([0, 3, 10, 20, 30, 40, 50, 50, 100, 150, 250, 299, 1, 0, 1], dtype=np.uint64)  # SURVEY 8c stream
    cum = np.cumsum(counts) / counts.sum()
    for pct in [0.0, 0.1, 0.25, 0.5, 0.75, 0.95, 0.99, 0.9999, 1.0]:
        want = next(i for i in range(15) if counts[i] and pct <= cum[i])
        assert L.gyo_slab_percentile_idx(oracle.ptr(counts, oracle.u64p), 15, pct) == want
    assert L.gyo_slab_percentile_idx(oracle.ptr(np.zeros(15, dtype=np.uint64), oracle.u64p), 15, 0.95) == 1  # empty -> bucket 1
    # get_stats: thresholds of those buckets, negative clamped to 0 (gy_statistics.h:1352-1356)
    h = oracle.MLHist()
    L.gyo_mlh_init(C.byref(h), oracle.RESP_TIME_HASH, NB)
    stats = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
    stats["count"][:15] = counts
    stats["sum"][:15] = counts.astype(np.int64) * 7
    L.gyo_mlh_add_hist(C.byref(h), 1000, stats.ctypes.data, 1)
    pcts = np.array([25, 50, 95, 99.99], dtype=np.float32)
    vals = np.zeros(4, dtype=np.int64)
    tc, ts, mean = C.c_int64(), C.c_int64(), C.c_double()
    L.gyo_mlh_get_stats(C.byref(h), 1, oracle.ptr(pcts, oracle.f32p), 4, oracle.ptr(vals, oracle.i64p), C.byref(tc), C.byref(ts), C.byref(mean))
    assert tc.value == int(counts.sum()) and ts.value == 7 * int(counts.sum()) and mean.value == 7.0
    # p99.99 of 1004 values lies in the overflow bucket under the slab rule (pct <= cumulative fraction); its ceiling for
    # RESP_TIME_HASH is SHRT_MAX (get_bucket_max_threshold gy_statistics.h:506-512)
    assert vals.tolist() == [300, 700, 1000, 32767]
    e = oracle.MLHist()
    L.gyo_mlh_init(C.byref(e), oracle.RESP_TIME_HASH, NB)
    L.gyo_mlh_get_stats(C.byref(e), 2, oracle.ptr(pcts, oracle.f32p), 4, oracle.ptr(vals, oracle.i64p), C.byref(tc), C.byref(ts), C.byref(mean))
    assert vals.tolist() == [1, 1, 1, 1] and tc.value == 0 and mean.value == 0.0


# ---------------------------------------------------------------- get_stats_for_period: count(start, end) / sum(start, end)
def _f32_scaled(v, num, den):
    """rangeAdjust: input * ((end - start) * 1.f / (bucket width)) in float, truncated back to the integer type"""
    scale = np.float32(np.float32(num) * np.float32(1.0) / np.float32(den))
    return int(np.float32(np.float32(v) * scale))


def brute_range(adds, latest, first, level, start, end):
    """a second statement of BucketedTimeSeries::count/sum(start, end) after update(latest): the adds grouped by the ring bucket that
    holds them, every bucket [bs, bn) scaled by its overlap with [start, end); the bucket holding `latest` ends at latest + 1"""
    dur = LEVEL_SECS[level]
    if dur == 0:
        spans = [(first, latest + 1)]
    else:
        nb = min(NB, dur)
        cur = (latest % dur) * nb // dur
        spans = []
        for back in range(nb - 1, -1, -1):  # oldest first
            j = cur - back
            cyc = latest // dur
            if j < 0:
                j += nb
                cyc -= 1
            bs = cyc * dur + -(-(j * dur) // nb)
            bn = cyc * dur + -(-((j + 1) * dur) // nb)
            spans.append((bs, bn))
    tc = ts = 0
    for bs, bn in spans:
        if start >= bn:
            continue
        if end <= bs:
            break
        c = sum(c for t, s, c in adds if bs <= t < bn)
        s = sum(s for t, s, c in adds if bs <= t < bn)
        if bs <= latest < bn:
            bn = latest + 1
        if start <= bs and end >= bn:
            tc, ts = tc + c, ts + s
        else:
            num, den = min(end, bn) - max(start, bs), bn - bs
            tc, ts = tc + _f32_scaled(c, num, den), ts + _f32_scaled(s, num, den)
    return tc, ts


@pytest.mark.parametrize("cadence", ["regular5", "jitter", "gaps"])
def test_period_range_matches_definition(cadence):
    L = oracle.lib()
    L.gyo_bts_range.argtypes = [C.POINTER(oracle.BTS), C.c_int64, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    L.gyo_bts_range.restype = None
    rng = np.random.default_rng(23)
    series = [oracle.BTS() for _ in range(4)]
    for lv, s in enumerate(series):
        L.gyo_bts_init(C.byref(s), NB, LEVEL_SECS[lv])
    t = 1_700_000_003
    adds, first = [], None
    for step in range(400):
        if cadence == "regular5":
            t += 5
        elif cadence == "jitter":
            t += int(rng.integers(1, 9))
        else:
            t += int(rng.choice([5, 5, 5, 40, 301, 3600, 43200 * 3, 5 * 24 * 3600 + 7]))
        if rng.random() < 0.8:
            sm, cnt = int(rng.integers(0, 10**9)), int(rng.integers(1, 10**6))
            adds.append((t, sm, cnt))
            for s in series:
                L.gyo_bts_add(C.byref(s), t, sm, cnt)
        for s in series:
            L.gyo_bts_update(C.byref(s), t)
        first = first if first is not None else t
        if step % 3:
            continue
        for lv, s in enumerate(series):
            dur = LEVEL_SECS[lv] or (t - first + 10)
            for _ in range(6):
                a = t - int(rng.integers(0, dur + dur // 4 + 2))
                b = a + int(rng.integers(1, dur + 2))
                c, sm_ = C.c_uint64(), C.c_int64()
                L.gyo_bts_range(C.byref(s), a, b, C.byref(c), C.byref(sm_))
                want = brute_range([x for x in adds if x[0] <= t], t, first, lv, a, b)
                # adds older than the ring are not in it: brute_range only looks inside the ring's spans, like the ring
                assert (c.value, sm_.value) == want, (cadence, step, lv, a - t, b - t)
            # the whole span of the level == its totals
            c, sm_ = C.c_uint64(), C.c_int64()
            L.gyo_bts_range(C.byref(s), 0, t + 1, C.byref(c), C.byref(sm_))
            assert (c.value, sm_.value) == (s.tot_cnt, s.tot_sum)


def test_period_level_choice_and_stats():
    """MultiLevelTimeSeries::getLevel(start): the first level whose duration reaches back to start; the percentile rule on the
    interval counts (CountFromInterval); a period that covers a level's whole ring gives that level's get_stats"""
    L = oracle.lib()
    h = oracle.MLHist()
    L.gyo_mlh_init(C.byref(h), oracle.RESP_TIME_HASH, NB)
    rng = np.random.default_rng(5)
    t = 1_700_000_000
    for w in range(200):
        t += 5
        st = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
        st["count"][:15] = rng.integers(0, 50, 15)
        st["sum"][:15] = st["count"][:15] * rng.integers(1, 2000, 15)
        L.gyo_mlh_add_hist(C.byref(h), t, st.ctypes.data, 1)
        L.gyo_mlh_flush(C.byref(h), t)
    assert [L.gyo_mlh_level_for_start(C.byref(h), 0, t - d) for d in (0, 5, 6, 300, 301, 432000, 432001, 10**8)] == [0, 0, 1, 1, 2, 2, 3, 3]
    p = np.array([25.0, 50.0, 95.0, 99.0], dtype=np.float32)
    for lv, back in ((1, 300), (2, 432000), (3, 10**8)):
        v1, v2 = np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64)
        a = [C.c_int64(), C.c_int64(), C.c_double()]
        b = [C.c_int64(), C.c_int64(), C.c_double()]
        L.gyo_mlh_get_stats(C.byref(h), lv, oracle.ptr(p, oracle.f32p), 4, oracle.ptr(v1, oracle.i64p), C.byref(a[0]), C.byref(a[1]), C.byref(a[2]))
        L.gyo_mlh_get_stats_for_period(C.byref(h), t - back, t, oracle.ptr(p, oracle.f32p), 4, oracle.ptr(v2, oracle.i64p), C.byref(b[0]), C.byref(b[1]),
                                       C.byref(b[2]))
        assert v1.tolist() == v2.tolist() and [x.value for x in a] == [x.value for x in b], lv
    # a sub-interval: half of a 30-s ring bucket of the 300-s level is scaled in float
    out = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
    L.gyo_mlh_period(C.byref(h), t - 100, t - 50, out.ctypes.data)
    full = np.zeros(16, dtype=oracle.HIST_SERIAL_DT)
    L.gyo_mlh_level(C.byref(h), 1, full.ctypes.data)
    assert 0 < int(out["count"].sum()) < int(full["count"].sum())
