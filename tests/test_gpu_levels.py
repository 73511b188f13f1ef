"""GPU parity for the multi-level windows: the engine answers "last 5 s / 300 s / 5 days / since start" from cumulative
snapshots taken at ring-bucket boundaries; the oracle (oracle/gy_oracle_levels.c) keeps folly-style rings per histogram bucket the way
TIME_HISTOGRAM does (common/gy_statistics.h:1082-1551).  Both must give the same {count, sum} per bucket and level at every window
close and at arbitrary query times in between, over regular, jittered and gapped close times (30-s, 300-s, 12-h and 5-day boundaries),
for the lazily rolled records (t-digest on) and the eagerly folded ones; plus LISTENER_DAY_STATS (common/gy_comm_proto.h:1620-1632)."""
import ctypes as C

import numpy as np
import pytest

from gyeeta_amd import wire
from tests import helpers

pytestmark = pytest.mark.gpu

NB = 10
T0 = 1_700_000_003


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    return torch


def _engine(**kw):
    from gyeeta_amd.engine import SketchEngine
    return SketchEngine(**kw)


class RingOracle:
    """one gyo_mlhist per service"""

    def __init__(self, oracle, nsvc):
        self.o, self.L, self.n = oracle, oracle.lib(), nsvc
        self.h = [oracle.MLHist() for _ in range(nsvc)]
        for h in self.h:
            self.L.gyo_mlh_init(C.byref(h), oracle.RESP_TIME_HASH, NB)

    def close(self, t, win_hist):
        """win_hist [nsvc][16][2] {count, sum}: add_histogram_data(t, ..., flush) for every service, then flush(t)"""
        for s, h in enumerate(self.h):
            st = np.zeros(16, dtype=self.o.HIST_SERIAL_DT)
            st["count"][:15] = win_hist[s, :15, 0]
            st["sum"][:15] = win_hist[s, :15, 1]
            self.L.gyo_mlh_add_hist(C.byref(h), t, st.ctypes.data, 1)
            self.L.gyo_mlh_flush(C.byref(h), t)

    def level(self, level, tq):
        """[nsvc][15][2] as of tq (on copies: a query must not disturb the series)"""
        out = np.zeros((self.n, 15, 2), dtype=np.int64)
        buf = np.zeros(16, dtype=self.o.HIST_SERIAL_DT)
        for s, h in enumerate(self.h):
            hc = self.o.MLHist.from_buffer_copy(h)
            self.L.gyo_mlh_flush(C.byref(hc), tq)
            self.L.gyo_mlh_level(C.byref(hc), level, buf.ctypes.data)
            out[s, :, 0] = buf["count"][:15].astype(np.int64)
            out[s, :, 1] = buf["sum"][:15]
        return out

    def period(self, a, b, tq):
        """[nsvc][15][2] of the seconds [a, b] as of tq, and the level folly answers from"""
        out = np.zeros((self.n, 15, 2), dtype=np.int64)
        buf = np.zeros(16, dtype=self.o.HIST_SERIAL_DT)
        lv = -1
        for s, h in enumerate(self.h):
            hc = self.o.MLHist.from_buffer_copy(h)
            self.L.gyo_mlh_flush(C.byref(hc), tq)
            self.L.gyo_mlh_period(C.byref(hc), a, b, buf.ctypes.data)
            out[s, :, 0] = buf["count"][:15].astype(np.int64)
            out[s, :, 1] = buf["sum"][:15]
            lv = self.L.gyo_mlh_level_for_start(C.byref(hc), 0, a)
        return out, lv

    def period_stats(self, s, a, b, tq, pcts):
        hc = self.o.MLHist.from_buffer_copy(self.h[s])
        self.L.gyo_mlh_flush(C.byref(hc), tq)
        p = np.array(pcts, dtype=np.float32)
        vals = np.zeros(len(pcts), dtype=np.int64)
        tc, ts, mean = C.c_int64(), C.c_int64(), C.c_double()
        self.L.gyo_mlh_get_stats_for_period(C.byref(hc), a, b, self.o.ptr(p, self.o.f32p), len(pcts), self.o.ptr(vals, self.o.i64p), C.byref(tc),
                                            C.byref(ts), C.byref(mean))
        return vals.tolist(), tc.value, ts.value, mean.value

    def stats(self, s, level, tq, pcts):
        hc = self.o.MLHist.from_buffer_copy(self.h[s])
        self.L.gyo_mlh_flush(C.byref(hc), tq)
        p = np.array(pcts, dtype=np.float32)
        vals = np.zeros(len(pcts), dtype=np.int64)
        tc, ts, mean = C.c_int64(), C.c_int64(), C.c_double()
        self.L.gyo_mlh_get_stats(C.byref(hc), level, self.o.ptr(p, self.o.f32p), len(pcts), self.o.ptr(vals, self.o.i64p), C.byref(tc), C.byref(ts),
                                 C.byref(mean))
        return vals.tolist(), tc.value, ts.value, mean.value


def _check_periods(eng, ring, rng, tq, nsvc, allmax, closes5=True):
    """get_stats_for_period at query time tq: intervals that end now / in the past / in the future, that lie inside one ring bucket,
    straddle several, reach past the ring, and pick each of the four levels"""
    spans = [(tq - 4, tq), (tq - 5, tq), (tq - 17, tq - 3), (tq - 60, tq), (tq - 299, tq - 31), (tq - 300, tq), (tq - 301, tq), (tq - 1000, tq - 200),
             (tq - 43200, tq), (tq - 100000, tq - 50000), (tq - 432000, tq + 10), (tq - 432001, tq), (tq - 10**7, tq - 100), (0, tq + 5),
             (tq + 3, tq + 9), (tq - 10**6, tq - 432000 - 50)]
    for _ in range(4):
        a = tq - int(rng.integers(0, 500000))
        spans.append((a, a + int(rng.integers(0, 500000))))
    for a, b in spans:
        g, lv = eng.export_hist_period(a, b, tq * 1_000_000, 0, nsvc)
        if lv == 0 and not closes5:
            continue  # level 0 is the engine's tumbling window: folly's 5-s ring only when closes are >= 5 s apart
        o, olv = ring.period(a, b, tq)
        assert lv == olv, (a - tq, b - tq, lv, olv)
        bad = np.argwhere(g[:, :15, :] != o)
        assert bad.size == 0, f"period [{a - tq}, {b - tq}] level {lv} at t={tq}: {bad[:4].tolist()} gpu {g[tuple(bad[0][:2])]} oracle {o[tuple(bad[0][:2])]}"
        assert (g[:, 15, 0] == o[:, :, 0].sum(axis=1)).all()
        assert (g[:, 15, 1] == allmax).all()


def _check_levels(eng, ring, tq, nsvc, levels, allmax):
    for lv in levels:
        g = eng.export_hist_level(lv, tq * 1_000_000, 0, nsvc)
        o = ring.level(lv, tq)
        bad = np.argwhere(g[:, :15, :] != o)
        assert bad.size == 0, f"level {lv} at t={tq}: mismatch at {bad[:4].tolist()} gpu {g[tuple(bad[0][:2])]} oracle {o[tuple(bad[0][:2])]}"
        assert (g[:, 15, 0] == o[:, :, 0].sum(axis=1)).all()   # total_count of the level
        assert (g[:, 15, 1] == allmax).all()                   # max_val_seen: all-time maximum on every level


@pytest.mark.parametrize("enable_td", [True, False], ids=["lazy", "eager"])
def test_levels_match_ring_oracle(torch_mod, oracle, enable_td):
    rng = np.random.default_rng(77)
    nh, sp = 2, 6
    nsvc = nh * sp
    eng = _engine(max_hosts=4, max_services=32, max_batch_events=1 << 14, enable_tdigest=enable_td, enable_levels=True)
    orc_win = oracle.OracleEngine(32, enable_td=False)  # cleared at every close: the closing window's histograms
    orc_all = oracle.OracleEngine(32, enable_td=False)
    info, gids = helpers.register_world(eng, orc_win, range(nh), sp)
    helpers.register_world(None, orc_all, range(nh), sp)
    ring = RingOracle(oracle, nsvc)
    day5 = 5 * 24 * 3600
    steps = ([5] * 9 + [3, 4, 7, 2, 9, 1] + [5] * 4 + [40, 5, 5, 301, 5, 5, 43200, 5, 5, 3 * 43200 + 17, 5, day5 - 3 * 43200, 5, 5, day5 + 7, 5, 5] +
             [30] * 12 + [5, 5])
    t = T0
    prev_dt = 100
    for w, dt in enumerate(steps):
        t += dt
        # events of the window: some hosts / services silent, some windows empty
        if w % 7 != 5:
            for h in range(nh):
                if rng.random() < 0.75:
                    ev = helpers.make_resp_events(rng, h, int(rng.integers(1, 400)), int(rng.integers(1, sp + 1)), lat_mu=2.0 + 0.1 * (w % 20))
                    eng.handle_resp_events(info[h][0], ev)
                    for o in (orc_win, orc_all):
                        o.resp_batch(ev.tobytes(), [info[h][1]], [0])
        win = np.array(orc_win.hist()[:nsvc])
        if w % 5 == 2:
            # a query while the window is still open sees closed windows only
            _check_levels(eng, ring, t - 1, nsvc, [1, 2, 3], np.array(orc_all.hist()[:nsvc])[:, 15, 1])
        if w % 6 == 1:
            # prepared-but-not-finished state: the closing window counts already
            from gyeeta_amd import capi
            capi.check(eng.L.gys_window_prepare(eng.h, t * 1_000_000))
            ring.close(t, win)
            _check_levels(eng, ring, t, nsvc, [1, 2, 3], np.array(orc_all.hist()[:nsvc])[:, 15, 1])
            capi.check(eng.L.gys_window_finish(eng.h))
        else:
            eng.window_close(t * 1_000_000)
            ring.close(t, win)
        allmax = np.array(orc_all.hist()[:nsvc])[:, 15, 1]
        # level 0 is the engine's tumbling window: equal to folly's 5-s ring whenever closes are at least 5 s apart
        lv0 = eng.export_hist_level(0, t * 1_000_000, 0, nsvc)
        assert (lv0[:, :15, :] == win[:, :15, :]).all()
        _check_levels(eng, ring, t, nsvc, ([0] if prev_dt >= 5 and dt >= 5 else []) + [1, 2, 3], allmax)
        for probe in (2, 5, 31, 299, 43201):  # later queries without a close in between
            if dt >= 5 and prev_dt >= 5 or probe >= 5:
                _check_levels(eng, ring, t + probe, nsvc, [0, 1, 2, 3], allmax)
        if w % 3 == 0 or dt > 40:
            _check_periods(eng, ring, rng, t, nsvc, allmax, closes5=prev_dt >= 5 and dt >= 5)
            _check_periods(eng, ring, rng, t + int(rng.integers(1, 400)), nsvc, allmax, closes5=prev_dt >= 5 and dt >= 5)
        if w % 4 == 0:
            s = int(rng.integers(0, nsvc))
            gid = int(gids[s // sp][s % sp])
            for a, b in ((t - 100, t), (t - 4000, t - 20), (t - 3, t), (0, t)):
                got = eng.query_hist_period_stats(gid, a, b, t * 1_000_000, [25.0, 50.0, 95.0, 99.0])
                want = ring.period_stats(s, a, b, t, [25.0, 50.0, 95.0, 99.0])
                if not (a == t - 3 and (prev_dt < 5 or dt < 5)):
                    assert got == want, (w, s, a - t, b - t, got, want)
            for lv in (1, 2, 3):
                got = eng.query_hist_level_stats(gid, lv, t * 1_000_000, [25.0, 50.0, 95.0, 99.0])
                want = ring.stats(s, lv, t, [25.0, 50.0, 95.0, 99.0])
                assert got == want, (w, s, lv, got, want)
        orc_win.window_clear(clear_hist=True)
        orc_all.window_clear(clear_hist=False)
        prev_dt = dt
    eng.close()


def test_levels_need_the_config_flag(torch_mod):
    from gyeeta_amd import capi
    eng = _engine(max_hosts=2, max_services=8, max_batch_events=1 << 10)
    helpers.register_world(eng, None, range(1), 4)
    with pytest.raises(capi.GysError) as ei:
        eng.export_hist_level(1, 0, 0, 4)
    assert ei.value.code == capi.ERR_STATE
    with pytest.raises(capi.GysError):
        eng.export_day_stats(0, 0, 4)
    eng.close()


def test_listener_day_stats(torch_mod, oracle):
    """QPS / active-connection histograms fed by the listener-state records, 5-day response level from the event stream"""
    rng = np.random.default_rng(5)
    L = oracle.lib()
    nh, sp = 2, 40
    nsvc = nh * sp
    eng = _engine(max_hosts=4, max_services=128, max_batch_events=1 << 15, enable_levels=True)
    orc_win = oracle.OracleEngine(128, enable_td=False)
    info, gids = helpers.register_world(eng, orc_win, range(nh), sp)
    ring = RingOracle(oracle, nsvc)
    qps = [oracle.Hist() for _ in range(nsvc)]
    act = [oracle.Hist() for _ in range(nsvc)]
    for s in range(nsvc):
        L.gyo_hist_init(C.byref(qps[s]), oracle.KINDS["SEMI_LOG_HASH_LO"])
        L.gyo_hist_init(C.byref(act[s]), oracle.KINDS["HASH_1_3000"])
    t = T0
    for w in range(40):
        t += 5 if w % 9 else 43200 + 5
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, 3000, sp, lat_mu=3.0 + 0.05 * w)
            eng.handle_resp_events(info[h][0], ev)
            orc_win.resp_batch(ev.tobytes(), [info[h][1]], [0])
            svc = np.sort(rng.choice(sp, size=int(rng.integers(1, sp + 1)), replace=False))
            rec = wire.synth_listener_states(rng, h, svc, delete_frac=0.02, bad_state_frac=0.02)
            eng.partha_listener_state(info[h][0], wire.pack_variable(rec, None), len(rec))
            for r in rec:
                if r["query_flags"] == wire.LISTEN_FLAG_DELETE or r["curr_state"] > 5:
                    continue  # neither a deleted listener nor an invalid record reaches set_state
                s = h * sp + int(np.where(gids[h] == r["glob_id"])[0][0])
                L.gyo_hist_add(C.byref(qps[s]), int(r["nqrys_5s"]) // 5)
                L.gyo_hist_add(C.byref(act[s]), int(r["nconns_active"]))
        eng.window_close(t * 1_000_000)
        ring.close(t, np.array(orc_win.hist()[:nsvc]))
        orc_win.window_clear(clear_hist=True)
    ds = eng.export_day_stats(t * 1_000_000, 0, nsvc)
    gq, ga = eng.export_svc_hist(0, 0, nsvc), eng.export_svc_hist(1, 0, nsvc)
    for s in range(nsvc):
        d = ds[s]
        assert d.glob_id == int(gids[s // sp][s % sp])
        vals, tc, ts, _ = ring.stats(s, 2, t, [95.0, 25.0])
        assert (d.tcount_5d, d.tsum_5d, d.p95_5d_respms, d.p25_5d_respms) == (tc, ts, vals[0], vals[1]), s
        for hist, g, got in ((qps[s], gq[s], (d.p95_qps, d.p25_qps)), (act[s], ga[s], (d.p95_nactive, d.p25_nactive))):
            assert g[:15, 0].tolist() == [hist.stats[i].count for i in range(15)]
            assert g[:15, 1].tolist() == [hist.stats[i].sum for i in range(15)]
            assert (g[15, 0], g[15, 1]) == (hist.total_count, hist.max_val_seen)
            pd = (oracle.HistData * 2)()
            pd[0].percentile, pd[1].percentile = 95.0, 25.0
            L.gyo_hist_percentiles(C.byref(hist), pd, 2, None, None, None)
            assert got == (pd[0].data_value & 0xFFFFFFFF, pd[1].data_value & 0xFFFFFFFF), s
    assert any(d.tcount_5d > 0 for d in ds) and any(d.p95_qps > 1 for d in ds)
    eng.close()
