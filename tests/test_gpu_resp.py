"""GPU parity for the response-event hot path (SURVEY 8a rows a1,a2,a3,a4,a7,a8 + HLL/CMS/t-digest): every register the HIP
kernels produce is compared bit-for-bit with the CPU oracle on the same events; t-digest quantiles are additionally checked
against an exact sort (rank error <= 1 %) and against GY_HISTOGRAM bucket ceilings."""
import ctypes as C

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    return torch


def _engine(**kw):
    from gyeeta_amd.engine import SketchEngine
    return SketchEngine(**kw)


def _compare_all(eng, orc, oracle, check_td=True):
    nsvc = orc.nsvc
    helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc.hist(), nsvc)
    assert (eng.export_conn_bitmap(0, nsvc) == orc.bitmap()).all()
    if check_td:
        gs, gc, gm = eng.export_tdigest(0, nsvc)
        os_, oc, om = orc.td_arrays()
        assert (gc == oc).all(), f"digest counts differ at {np.argwhere(gc != oc)[:4].tolist()}"
        assert (gs == os_).all()
        assert (gm == om).all()
        gn, gp = eng.export_tdigest_pending(0, nsvc)  # buffered values (compared as sorted multisets)
        on, op = orc.td_pending()
        assert (gn == on).all(), f"buffer fill differs at {np.argwhere(gn != on)[:4].tolist()}"
        assert (gp == op).all()


def _compare_window(eng, orc):
    assert (eng.export_hll() == orc.hll()).all()
    assert (eng.export_cms(0) == orc.cms()).all()
    gh = eng.export_global_hist()
    oh, omax = orc.ghist()
    assert [gh.stats[i].count for i in range(15)] == oh[:15, 0].tolist()
    assert [gh.stats[i].sum for i in range(15)] == oh[:15, 1].tolist()
    assert gh.total_count == oh[15, 0] and gh.max_val_seen == omax


PATHS = pytest.mark.parametrize("resp_path", [1, 2], ids=["general", "hostlocal"])


def _assert_path(eng, resp_path):
    c = eng.counters()
    if resp_path == 1:
        assert c["resp_batches_host_local"] == 0 and c["resp_batches_general"] > 0
    elif resp_path == 2:
        assert c["resp_batches_general"] == 0 and c["resp_batches_host_local"] > 0


@PATHS
def test_resp_small_hosts_edge_cases(torch_mod, oracle, resp_path):
    rng = np.random.default_rng(1)
    nh, sp = 3, 7
    eng = _engine(max_hosts=8, max_services=64, max_batch_events=1 << 16, resp_path=resp_path)
    orc = oracle.OracleEngine(64)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    total = 0
    for rnd in range(4):
        for h in range(nh):
            n = int(rng.integers(1, 3000))
            ev = helpers.make_resp_events(rng, h, n, sp)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
            total += n
    eng.handle_resp_events(info[0][0], np.zeros(0, dtype=helpers.wire.RESP_EVENT))  # empty batch is a no-op
    eng.sync()
    _compare_all(eng, orc, oracle)
    _assert_path(eng, resp_path)
    c, oc = eng.counters(), orc.counters()
    assert c["resp_events"] == total == oc["events"]
    assert c["resp_dropped_range"] == oc["dropped_range"] and c["resp_dropped_nolistener"] == oc["dropped_nolistener"]
    # percentiles / quantiles of every service vs the oracle (bit exact) and vs GY_HISTOGRAM semantics
    hist = eng.export_hist(0, 0, orc.nsvc)
    for h in range(nh):
        for s in range(sp):
            g = int(gids[h][s])
            slot = eng.lookup(g)
            vals, sums, counts, tot, mx, avg = eng.hist_percentiles(g, [25.0, 50.0, 95.0, 99.0, 99.99], which=0)
            ov, os_, ocn, oavg = oracle.hist_percentiles(0, hist[slot][:15], hist[slot][15][0], [25.0, 50.0, 95.0, 99.0, 99.99])
            assert vals == ov and sums == os_ and counts == ocn
            assert np.float32(avg).tobytes() == np.float32(oavg).tobytes()
            qs = [0.0, 0.01, 0.25, 0.5, 0.9, 0.99, 1.0]
            gq = eng.quantiles(g, qs)
            oq = [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(slot)), q) for q in qs]
            assert gq == oq
    # window close: HLL / CMS / global histogram registers bit exact; histograms fold into the all-time set
    eng.window_close()
    _compare_window(eng, orc)
    helpers.assert_hist_equal(eng.export_hist(1, 0, orc.nsvc), orc.hist(), orc.nsvc)
    assert eng.export_hist(0, 0, orc.nsvc)[:, :15].sum() == 0  # window cleared
    est = eng.distinct_flows()
    assert est == pytest.approx(oracle.lib().gyo_hll_estimate(oracle.ptr(orc.hll(), oracle.u8p), 14), rel=1e-12)
    for h in range(nh):
        g = int(gids[h][0])
        w = oracle.glob_id_words(g)
        assert eng.cms(g, 0) == oracle.lib().gyo_cms_query(oracle.ptr(orc.cms(), oracle.u32p), oracle.ptr(w, oracle.u32p), 2)
    eng.close()


@PATHS
def test_resp_device_generated_stream(torch_mod, oracle, resp_path):
    """SURVEY 8d-style stream generated ON the GPU, replayed through the oracle on the host from the very same bytes"""
    torch = torch_mod
    nh, sp, n = 32, 50, 1 << 19
    eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=n, resp_path=resp_path)
    orc = oracle.OracleEngine(nh * sp)
    helpers.register_world(eng, orc, range(nh), sp)
    ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    exact = {}
    for rnd in range(3):
        segs = eng.gen_resp_events(ev.data_ptr(), n, 1234 + rnd, 0, nh, sp)
        eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
        eng.sync()
        host = ev.cpu().numpy().tobytes()
        orc.resp_batch(host, [s.host_slot for s in segs], [s.first_event for s in segs])
        a = np.frombuffer(host, dtype=helpers.wire.RESP_EVENT)
        exact[rnd] = a
    eng.sync()
    _compare_all(eng, orc, oracle)
    _assert_path(eng, resp_path)
    assert eng.counters()["resp_dropped_nolistener"] == 0
    # t-digest vs exact sort: rank error <= 1 % (north_star tolerance), and bucket agreement with GY_HISTOGRAM::get_percentile
    allev = np.concatenate([exact[r] for r in range(3)])
    lat = (allev["lsndtime"] - allev["lrcvtime"]).astype(np.int64)
    hostidx = np.repeat(np.arange(nh), n // nh)
    hostidx = np.concatenate([hostidx] * 3)
    port = allev["sport_be"].astype(np.int64)
    worst = 0.0
    for h in (0, 7, 31):
        for s in (0, 13, 49):
            sel = (hostidx == h) & (port == 1024 + s)
            x = np.sort(lat[sel])
            g = int(helpers.wire.glob_id(h, s))
            for q, gq in zip((0.5, 0.99), eng.quantiles(g, [0.5, 0.99])):
                lo = np.searchsorted(x, gq, side="left") / len(x)
                hi = np.searchsorted(x, gq, side="right") / len(x)
                err = 0.0 if lo <= q <= hi else min(abs(lo - q), abs(hi - q))
                worst = max(worst, err)
    assert worst <= 0.01, f"t-digest rank error {worst:.4f} > 1%"
    eng.window_close()
    _compare_window(eng, orc)
    eng.close()


@PATHS
@pytest.mark.parametrize("td_buf", [0, 1024], ids=["bufauto", "buf1024"])
def test_resp_huge_key_batches(torch_mod, oracle, resp_path, td_buf):
    """keys whose batch does not fit their value buffer are spilled (second resp pass / run in the staging area) and, above 16384
    values, take the value-count-array kernel; the result must still equal the oracle, including when a small batch, a huge batch and
    another small batch hit the same key"""
    rng = np.random.default_rng(5)
    eng = _engine(max_hosts=2, max_services=8, max_batch_events=1 << 18, resp_path=resp_path, td_buf_values=td_buf)
    orc = oracle.OracleEngine(8)
    info, gids = helpers.register_world(eng, orc, range(1), 3)
    mid, slot = info[0]
    for n, sp, mu in [(300, 3, 3.0), (150000, 1, 4.0), (900, 3, 2.0), (200000, 2, 6.5), (1, 1, 1.0), (1025, 1, 3.0), (1024, 1, 3.0)]:
        ev = helpers.make_resp_events(rng, 0, n, sp, lat_mu=mu, bad_frac=0.01, unknown_frac=0.0)
        eng.handle_resp_events(mid, ev)
        orc.resp_batch(ev.tobytes(), [slot], [0])
    eng.sync()
    _compare_all(eng, orc, oracle)
    _assert_path(eng, resp_path)
    for s in range(3):
        g = int(gids[0][s])
        qs = [0.001, 0.5, 0.999]
        assert eng.quantiles(g, qs) == [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(s)), q) for q in qs]
    eng.close()


@PATHS
@pytest.mark.parametrize("td_cap", [0, 1920], ids=["cap896", "cap1920"])
def test_resp_merge_size_classes(torch_mod, oracle, resp_path, td_cap):
    """one key, one batch per merge size class and on each side of the class boundaries (round 6: 2 049 .. 4 096 values go through
    k_digest_bins<false,16>, 4 097 .. 16 384 through the streamed instance k_digest_bins<false,64>, more through the several-workgroup
    path): every batch is larger than the buffer, so every call ends in a merge of exactly its own values.  Batches of slow responses
    (mu 6.5 / 7.5: thousands of values of a second or longer) are what the value-bin instances hand over to the general kernel
    (k_digest_merge<4096,256> / <16384,1024>)."""
    rng = np.random.default_rng(61)
    eng = _engine(max_hosts=2, max_services=8, max_batch_events=1 << 16, resp_path=resp_path, td_pend_cap=td_cap)
    orc = oracle.OracleEngine(8, td_cap=td_cap)
    info, gids = helpers.register_world(eng, orc, range(1), 2)
    mid, slot = info[0]
    for n, mu in [(2049, 3.0), (4096, 3.0), (4097, 3.0), (9000, 3.0), (3000, 6.5), (12000, 6.5), (16384, 2.0), (16385, 2.0), (7000, 7.5), (5000, 0.5), (40000, 3.0), (4500, 3.0)]:
        ev = helpers.make_resp_events(rng, 0, n, 1, lat_mu=mu, bad_frac=0.0, unknown_frac=0.0, zero_ip_frac=0.0)
        eng.handle_resp_events(mid, ev)
        orc.resp_batch(ev.tobytes(), [slot], [0])
        eng.sync()
        _compare_all(eng, orc, oracle)
    _assert_path(eng, resp_path)
    g = int(gids[0][0])
    qs = [0.001, 0.25, 0.5, 0.95, 0.999]
    assert eng.quantiles(g, qs) == [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(0)), q) for q in qs]
    eng.close()


def test_tdigest_disabled_and_identical_values(torch_mod, oracle):
    rng = np.random.default_rng(9)
    eng = _engine(max_hosts=1, max_services=4, enable_tdigest=False)
    orc = oracle.OracleEngine(4, enable_td=False)
    info, gids = helpers.register_world(eng, orc, range(1), 2)
    ev = helpers.make_resp_events(rng, 0, 5000, 2)
    eng.handle_resp_events(info[0][0], ev)
    orc.resp_batch(ev.tobytes(), [0], [0])
    eng.sync()
    _compare_all(eng, orc, oracle, check_td=False)
    from gyeeta_amd import capi
    with pytest.raises(capi.GysError):
        eng.quantiles(int(gids[0][0]), [0.5])
    eng.close()
    # all-identical values (every item ties with every other): exercises the tie rule old-before-new
    eng = _engine(max_hosts=1, max_services=4, max_batch_events=1 << 16, resp_path=2)
    orc = oracle.OracleEngine(4)
    info, gids = helpers.register_world(eng, orc, range(1), 1)
    for n in (10, 2000, 700):
        ev = helpers.make_resp_events(rng, 0, n, 1, bad_frac=0, unknown_frac=0, zero_ip_frac=0)
        ev["lsndtime"] = ev["lrcvtime"] + np.uint32(42)
        eng.handle_resp_events(info[0][0], ev)
        orc.resp_batch(ev.tobytes(), [0], [0])
    eng.sync()
    _compare_all(eng, orc, oracle)
    assert eng.quantiles(int(gids[0][0]), [0.1, 0.5, 0.9]) == [42.0, 42.0, 42.0]
    eng.close()


def test_resp_hostlocal_incremental_registration_and_fallbacks(torch_mod, oracle):
    """listeners registered in several interleaved calls (non-contiguous slots, growing sub-tables), a host with no listeners, and a
    multi-segment batch that names one host twice (must take the general pipeline even when host-local is preferred)"""
    torch = torch_mod
    rng = np.random.default_rng(11)
    nh, sp = 4, 40
    eng = _engine(max_hosts=8, max_services=1024, max_batch_events=1 << 16, resp_path=2)
    orc = oracle.OracleEngine(1024)
    mids, slots = {}, {}
    for h in range(nh + 1):  # host nh never gets a listener
        mids[h] = helpers.wire.machine_id(h)
        slots[h] = eng.register_host(mids[h], "c")
    step = 5
    for lo in range(0, sp, step):  # interleave hosts so that a host's slots are scattered
        for h in range(nh):
            s = np.arange(lo, min(sp, lo + step))
            g = helpers.wire.glob_id(np.full(len(s), h), s)
            ns, pt = helpers.wire.listener_netns(h, s), helpers.wire.listener_port(s)
            eng.register_listeners_np(mids[h], g, ns, pt)
            for i in range(len(s)):
                orc.register(slots[h], int(g[i]), int(ns[i]), int(pt[i]))
        # ingest between registration rounds: the sub-tables must be valid at every size
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, 700, min(sp, lo + step + 3))  # some events name listeners not registered yet
            eng.handle_resp_events(mids[h], ev)
            orc.resp_batch(ev.tobytes(), [slots[h]], [0])
    ev = helpers.make_resp_events(rng, nh, 100, 3)
    eng.handle_resp_events(mids[nh], ev)  # host without listeners: every in-range event is dropped "no listener"
    orc.resp_batch(ev.tobytes(), [slots[nh]], [0])
    eng.sync()
    _compare_all(eng, orc, oracle)
    c0 = eng.counters()
    assert c0["resp_batches_general"] == 0 and c0["resp_dropped_nolistener"] == orc.counters()["dropped_nolistener"] > 0
    # multi-segment device batch: distinct hosts -> host-local; a repeated host -> general
    parts = [helpers.make_resp_events(rng, h, 500 + 37 * h, sp) for h in (0, 1, 2, 3)]
    buf = helpers.concat_events(parts)
    firsts = np.cumsum([0] + [len(x) for x in parts[:-1]])
    from gyeeta_amd import capi
    def run(hosts):
        segs = (capi.RespSeg * len(hosts))()
        for i, h in enumerate(hosts):
            segs[i].host_slot, segs[i].first_event = slots[h], int(firsts[i])
        d = torch.from_numpy(buf.view(np.uint8).copy()).cuda()
        eng.handle_resp_events_dev(segs, d.data_ptr(), len(buf))
        eng.sync()
        orc.resp_batch(buf.tobytes(), [slots[h] for h in hosts], [int(f) for f in firsts])
    run([0, 1, 2, 3])
    c1 = eng.counters()
    assert c1["resp_batches_host_local"] == c0["resp_batches_host_local"] + 1 and c1["resp_batches_general"] == 0
    # same bytes, but segment 2 is attributed to host 0 again: events whose (netns, port) belong to host 2 miss in host 0's table
    run([0, 1, 0, 3])
    c2 = eng.counters()
    assert c2["resp_batches_general"] == 1
    _compare_all(eng, orc, oracle)
    eng.window_close()
    _compare_window(eng, orc)
    eng.close()


@PATHS
@pytest.mark.parametrize("td_buf", [0, 1024], ids=["bufauto", "buf1024"])
def test_resp_buffer_fill_and_merge_cycles(torch_mod, oracle, resp_path, td_buf):
    """many small batches per key: the t-digest buffer fills (append), overflows (one merge of buffer + batch) and refills; batch
    sizes straddle the buffer capacity, the 64-lane chunk size of the per-key pass and the merge kernel's sort sizes"""
    rng = np.random.default_rng(21)
    nh, sp = 2, 6
    eng = _engine(max_hosts=2, max_services=16, max_batch_events=1 << 17, resp_path=resp_path, td_buf_values=td_buf)
    orc = oracle.OracleEngine(16)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    # (with a 1024-entry buffer the larger batches do not fit behind the buffered values: spilled keys, merged from buffer + run)
    sizes = [5, 40, 64, 65, 130, 255, 256, 257, 1, 700, 1024 * sp, 300, 3, 511, 9, 9, 9, 2000, 77, 767, 1, 768 * sp, 2, 4500 * sp, 20000 * sp, 30]
    for rnd, n in enumerate(sizes):
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, n, sp if rnd % 3 else 1, lat_mu=2.0 + 0.2 * rnd, bad_frac=0.01, unknown_frac=0.01)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
        if rnd % 5 == 4:
            eng.sync()
            _compare_all(eng, orc, oracle)
            for h in range(nh):
                g = int(gids[h][0])
                qs = [0.0, 0.1, 0.5, 0.95, 1.0]
                assert eng.quantiles(g, qs) == [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(eng.lookup(g))), q) for q in qs]
    eng.sync()
    _compare_all(eng, orc, oracle)
    _assert_path(eng, resp_path)
    npend, _ = eng.export_tdigest_pending(0, orc.nsvc)
    from gyeeta_amd import capi
    assert npend.max() <= capi.TD_PEND_CAP
    eng.window_close()
    _compare_window(eng, orc)
    eng.close()


@PATHS
@pytest.mark.parametrize("td_buf", [0, 1024], ids=["bufauto", "buf1024"])
def test_resp_windows_roll_lazily(torch_mod, oracle, resp_path, td_buf):
    """several 5-s windows in which some hosts / services stay silent: the window view, the all-time view, the CONN_BITMAP rows and
    the per-window registers must equal an oracle that clears / folds eagerly at every boundary (the engine rolls a key only when a
    later window touches it), including several batches per window and a window with no events at all"""
    rng = np.random.default_rng(31)
    nh, sp = 3, 9
    eng = _engine(max_hosts=4, max_services=64, max_batch_events=1 << 16, resp_path=resp_path, td_buf_values=td_buf)
    orc_all = oracle.OracleEngine(64)   # histograms never cleared  -> all-time view
    orc_win = oracle.OracleEngine(64)   # histograms cleared at every boundary -> window view
    info, gids = helpers.register_world(eng, orc_all, range(nh), sp)
    helpers.register_world(None, orc_win, range(nh), sp)
    pcts = np.array([50.0, 95.0, 99.0], dtype=np.float32)
    plan = [  # per window: list of (host, events, services touched)
        [(0, 900, sp), (1, 500, sp), (2, 40, 2)],
        [(0, 300, 3)],                                  # hosts 1, 2 silent; host 0 touches 3 of 9 services
        [],                                             # empty window
        [(1, 700, sp), (1, 200, 4), (0, 50, 1)],        # two batches of host 1 in one window
        [(2, 1200, sp), (0, 10, sp)],
        [(0, 9000, sp), (2, 3000, 2), (0, 2500, 1)],    # buffers overflow mid-window: merges fold the window's values, later batches add to it
        [(2, 2, 2)],
    ]
    for wnd, batches in enumerate(plan):
        for h, n, ns in batches:
            ev = helpers.make_resp_events(rng, h, n, ns, lat_mu=2.5 + 0.3 * wnd)
            eng.handle_resp_events(info[h][0], ev)
            for o in (orc_all, orc_win):
                o.resp_batch(ev.tobytes(), [info[h][1]], [0])
        eng.sync()
        nsvc = orc_all.nsvc
        helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc_win.hist(), nsvc)   # window view
        helpers.assert_hist_equal(eng.export_hist(1, 0, nsvc), orc_all.hist(), nsvc)   # all-time view (folded + unfolded)
        assert (eng.export_conn_bitmap(0, nsvc) == orc_win.bitmap()).all()
        # the device-side per-key percentile scan sees the same views
        torch = torch_mod
        for which, o in ((0, orc_win), (1, orc_all)):
            d_out = torch.zeros(nsvc * len(pcts), dtype=torch.int64, device="cuda")
            from gyeeta_amd import capi
            eng.order()
            capi.check(eng.L.gys_scan_percentiles_dev(eng.h, which, pcts.ctypes.data_as(capi.f32p), len(pcts), C.c_void_p(d_out.data_ptr())))
            eng.sync()
            got = d_out.cpu().numpy().reshape(nsvc, len(pcts))
            oh = o.hist()
            for s in range(nsvc):
                ov, _, _, _ = oracle.hist_percentiles(0, oh[s][:15], oh[s][15][0], [float(x) for x in pcts])
                assert got[s].tolist() == ov
        eng.window_close()
        _compare_window(eng, orc_win)
        for o in (orc_all, orc_win):
            o.window_clear(clear_hist=o is orc_win)
    _assert_path(eng, resp_path)
    assert eng.counters()["window_graph_launches"] == len(plan)  # every boundary replayed the captured hipGraph
    eng.close()


@pytest.mark.parametrize("td_buf", [0, 2048], ids=["bufauto", "buf2048"])
def test_resp_split_form_few_hosts_long_segments(torch_mod, oracle, td_buf):
    """few hosts with long segments: the host-local pipeline in its split form (parts of 65536 events: per-part counts, per-host scan,
    per-part scatter) must leave exactly what the fused form leaves -- every register vs the oracle over several batches and a window
    roll; segment lengths straddle the part size (one part, just over one part, several parts with a short tail), one host has a
    single listener, one batch is short enough to stay fused"""
    rng = np.random.default_rng(91)
    svc = {0: 37, 1: 1, 2: 200, 3: 64}
    eng = _engine(max_hosts=4, max_services=512, max_batch_events=1 << 20, resp_path=3, td_buf_values=td_buf)
    orc = oracle.OracleEngine(512)
    info = {}
    for h, sp in svc.items():
        i, _ = helpers.register_world(eng, orc, [h], sp)
        info.update(i)
    plans = [
        [(0, 200_000), (1, 65_537), (2, 30_000), (3, 65_536)],
        [(2, 400_001), (0, 70_000)],
        [(3, 10_000), (1, 20_000)],           # no segment longer than a part: fused form
        [(1, 131_072), (3, 300_000), (0, 1)],
    ]
    for b, plan in enumerate(plans):
        parts, seg_host, seg_first, pos = [], [], [], 0
        for h, n in plan:
            parts.append(helpers.make_resp_events(rng, h, n, svc[h], lat_mu=2.5 + 0.4 * b))
            seg_host.append(info[h][1])
            seg_first.append(pos)
            pos += n
        buf = helpers.concat_events(parts)
        d = torch_mod.from_numpy(buf.view(np.uint8).copy()).cuda()
        eng.order()
        from gyeeta_amd import capi
        segs = (capi.RespSeg * len(plan))()
        for i in range(len(plan)):
            segs[i].host_slot, segs[i].first_event = seg_host[i], seg_first[i]
        eng.handle_resp_events_dev(segs, d.data_ptr(), pos)
        eng.sync()
        orc.resp_batch(buf.tobytes(), seg_host, seg_first)
        _compare_all(eng, orc, oracle)
        if b == 1:
            eng.window_close()
            _compare_window(eng, orc)
            orc.window_clear(clear_hist=True)  # _compare_all looks at the window view
    c = eng.counters()
    assert (c["resp_batches_host_split"], c["resp_batches_host_local"], c["resp_batches_general"]) == (3, 1, 0)
    assert c["resp_events"] == orc.counters()["events"] and c["resp_dropped_nolistener"] == orc.counters()["dropped_nolistener"]
    eng.close()


@PATHS
def test_resp_ragged_segments_and_bad_arguments(torch_mod, oracle, resp_path):
    """multi-host device batches with empty segments (first, middle, last), a host without events, a batch of zero events, and the
    argument errors: first segment not at event 0, descending offsets, unknown host slot, batch above max_batch_events"""
    torch = torch_mod
    from gyeeta_amd import capi
    rng = np.random.default_rng(41)
    nh, sp = 5, 6
    eng = _engine(max_hosts=8, max_services=64, max_batch_events=1 << 14, resp_path=resp_path)
    orc = oracle.OracleEngine(64)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    slot = {h: info[h][1] for h in range(nh)}

    def run(lengths):  # lengths per host 0..nh-1 (0 = empty segment)
        parts = [helpers.make_resp_events(rng, h, n, sp) for h, n in enumerate(lengths)]
        buf = helpers.concat_events(parts)
        firsts = np.cumsum([0] + list(lengths[:-1]))
        segs = (capi.RespSeg * nh)()
        for h in range(nh):
            segs[h].host_slot, segs[h].first_event = slot[h], int(firsts[h])
        d = torch.from_numpy(np.frombuffer(buf.tobytes() + b"\0" * 24, dtype=np.uint8).copy()).cuda()
        eng.handle_resp_events_dev(segs, d.data_ptr(), len(buf))
        eng.sync()
        orc.resp_batch(buf.tobytes(), [slot[h] for h in range(nh)], [int(f) for f in firsts])

    run([0, 700, 0, 0, 300])      # empty first / middle segments
    run([50, 0, 9, 1, 0])         # empty last segment, tiny segments
    run([0, 0, 0, 0, 0])          # no events at all
    run([3000, 1, 2000, 0, 64])
    _compare_all(eng, orc, oracle)
    c, oc = eng.counters(), orc.counters()
    assert c["resp_events"] == oc["events"] and c["resp_dropped_nolistener"] == oc["dropped_nolistener"]
    # argument errors leave the state untouched
    d = torch.zeros(24 * 16, dtype=torch.uint8, device="cuda")
    segs = (capi.RespSeg * 2)()
    for bad in ([(slot[0], 1), (slot[1], 8)],          # first segment does not start at event 0
                [(slot[0], 0), (77, 8)]):              # unknown host slot
        for i, (hs, fe) in enumerate(bad):
            segs[i].host_slot, segs[i].first_event = hs, fe
        with pytest.raises(capi.GysError) as ei:
            eng.handle_resp_events_dev(segs, d.data_ptr(), 16)
        assert ei.value.code == capi.ERR_INVAL
    segs2 = (capi.RespSeg * 3)()
    for i, (hs, fe) in enumerate([(slot[0], 0), (slot[1], 9), (slot[2], 4)]):  # descending offsets
        segs2[i].host_slot, segs2[i].first_event = hs, fe
    with pytest.raises(capi.GysError):
        eng.handle_resp_events_dev(segs2, d.data_ptr(), 16)
    big = torch.zeros(24 * ((1 << 14) + 1), dtype=torch.uint8, device="cuda")
    one = (capi.RespSeg * 1)()
    one[0].host_slot = slot[0]
    with pytest.raises(capi.GysError) as ei:
        eng.handle_resp_events_dev(one, big.data_ptr(), (1 << 14) + 1)
    assert ei.value.code == capi.ERR_NOMEM
    eng.sync()
    _compare_all(eng, orc, oracle)
    eng.close()
