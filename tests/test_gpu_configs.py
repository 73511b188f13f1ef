"""BASELINE.json's configurations as GPU parity cases at their FULL sizes.

configs[0] (C1): 1 host x 100 services, long replay (the reference's own CPU-runnable case, here GPU vs the C oracle)
configs[1] (C2): 1 000 hosts x 100 services, TCP_CONN_NOTIFY flow stream + LISTENER_STATE_NOTIFY roll-up
configs[2] (C3): 10 000 hosts x 1 000 services = 10^7 service keys, response-event stream
configs[4] (C5): 10^5 services, Zipf(1.1) -- heavy hitters

Where the C oracle finishes in seconds (C2: 2^20 records, C5: 2^22 events) every register is compared bit for bit; at C3's size
the check is through size-independent properties: checksums of checksums (per-key totals vs the all-service histogram vs the event
counters vs the Count-Min row sums vs the digests' own totals), linearity of the window registers (HLL = max, CMS / histogram = sum
over disjoint batches), sortedness of the digest clusters, idempotence of queries."""
import ctypes as C

import numpy as np
import pytest

from gyeeta_amd import wire
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


def capi_default_cap():
    from gyeeta_amd import capi
    return capi.TD_PEND_CAP


def _register_bulk(eng, nhosts, svcs):
    s = np.arange(svcs)
    mids = []
    for h in range(nhosts):
        mid = wire.machine_id(h)
        assert eng.register_host(mid, "cluster%d" % (h % 4)) == h
        eng.register_listeners_np(mid, wire.glob_id(np.full(svcs, h), s), wire.listener_netns(h, s), wire.listener_port(s))
        mids.append(mid)
    return mids


# ---------------------------------------------------------------------------------------------------------------- C3
BENCH_TD_PEND_CAP = 1920  # bench.py's --td-pend-cap default: the headline configuration's gys_config (the library default is 896)
TD_CAPS = pytest.mark.parametrize("td_cap", [0, BENCH_TD_PEND_CAP], ids=["cap896-library-default", "cap1920-bench-default"])


@TD_CAPS
def test_c3_full_size_properties(torch_mod, oracle, td_cap):
    """td_cap 1920: the engine is created with exactly the gys_config bench.py's default run uses (10 000 hosts x 1 000 services, t-digest on,
    td_pend_cap 1920) and the 50-host oracle slice runs with the same buffer size."""
    torch = torch_mod
    nh, sp, n = 10_000, 1_000, 1 << 26
    nsvc = nh * sp
    eng = _engine(max_hosts=nh, max_services=nsvc, max_batch_events=n, td_pend_cap=td_cap)
    assert eng.L.gys_td_pend_cap(eng.h) == (td_cap or capi_default_cap())
    _register_bulk(eng, nh, sp)
    # bit-exact slice inside the full-size engine: the oracle is fed the whole segments of 50 hosts (the first and the last 25 host
    # slots = 50 000 of the 10^7 keys) of every batch the engine ingests, cut out of the very same bytes
    SLICE = list(range(25)) + list(range(nh - 25, nh))
    orc = oracle.OracleEngine(len(SLICE) * sp, td_cap=td_cap)
    for j, h in enumerate(SLICE):
        s = np.arange(sp)
        g, ns, pt = wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s)
        for i in range(sp):
            orc.register(j, int(g[i]), int(ns[i]), int(pt[i]))

    def oracle_feed(ev_dev, sg, nev):
        for j, h in enumerate(SLICE):
            idx = next((k for k, x in enumerate(sg) if x.host_slot == h), None)
            if idx is None:
                continue
            lo = sg[idx].first_event
            hi = sg[idx + 1].first_event if idx + 1 < len(sg) else nev
            orc.resp_batch(ev_dev[lo * 24:hi * 24].cpu().numpy().tobytes(), [j], [0])

    def slice_compare(window_open):
        for part, first in ((0, 0), (1, nsvc - 25 * sp)):
            k0, k1 = part * 25 * sp, (part + 1) * 25 * sp
            helpers.assert_hist_equal(eng.export_hist(1, first, 25 * sp), orc.hist()[k0:k1], 25 * sp)
            if window_open:
                assert (eng.export_conn_bitmap(first, 25 * sp) == orc.bitmap()[k0:k1]).all()
            gs, gc, gm = eng.export_tdigest(first, 25 * sp)
            os_, oc, om = orc.td_arrays()
            assert (gc == oc[k0:k1]).all() and (gs == os_[k0:k1]).all() and (gm == om[k0:k1]).all()
            gn, gp = eng.export_tdigest_pending(first, 25 * sp)
            on, op = orc.td_pending()
            assert (gn == on[k0:k1]).all() and (gp == op[k0:k1]).all()
    bufs, segs = [], []
    for b in range(2):
        ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        segs.append(eng.gen_resp_events(ev.data_ptr(), n, 0xC3 + b, 0, nh, sp))
        bufs.append(ev)
    eng.sync()

    def window(batches):
        for b in batches:
            eng.handle_resp_events_dev(segs[b], bufs[b].data_ptr(), n)
            oracle_feed(bufs[b], segs[b], n)
        slice_compare(True)
        eng.window_close()
        orc.window_clear(clear_hist=False)  # the oracle's record is the cumulative one: compared with the engine's all-time view
        gh = eng.export_global_hist()
        return (eng.export_hll().copy(), eng.export_cms(0).copy(), np.array([[gh.stats[i].count, gh.stats[i].sum] for i in range(15)], dtype=np.int64),
                gh.total_count, gh.max_val_seen)

    hll_a, cms_a, gh_a, tot_a, max_a = window([0])
    hll_b, cms_b, gh_b, tot_b, max_b = window([1])
    hll_ab, cms_ab, gh_ab, tot_ab, max_ab = window([0, 1])
    c = eng.counters()
    assert c["resp_events"] == 4 * n and c["resp_dropped_range"] == 0 and c["resp_dropped_nolistener"] == 0
    assert c["resp_batches_host_local"] == 4 and c["resp_batches_general"] == 0
    # linearity of the window registers over disjoint batches: HLL = max, Count-Min / histogram = sum, max = max
    assert (hll_ab == np.maximum(hll_a, hll_b)).all()
    assert (cms_ab == cms_a + cms_b).all()
    assert (gh_ab == gh_a + gh_b).all() and tot_ab == tot_a + tot_b == 2 * n and max_ab == max(max_a, max_b)
    # every Count-Min row counts every accepted event exactly once
    assert cms_a.sum(axis=1).tolist() == [n] * 4
    # all-service histogram of a window == sum of its buckets
    assert gh_a[:, 0].sum() == tot_a == n
    # distinct flows: clients are uniform in 10/8 x 49536 ports, so (almost) every event is a new flow; p = 14 -> 0.8 % std error
    est = eng.distinct_flows()
    assert abs(est - 2 * n) / (2 * n) < 0.04

    # ---- checksum of checksums over all 10^7 keys (all-time view, 4 batches), in slices of 10^6 keys
    tot_cnt = np.zeros(15, dtype=np.int64)
    tot_sum = np.zeros(15, dtype=np.int64)
    total = 0
    vmax = -1
    step = 1_000_000
    dig_total = 0
    dig_sum = 0
    for first in range(0, nsvc, step):
        h = eng.export_hist(1, first, step)
        tot_cnt += h[:, :15, 0].sum(axis=0)
        tot_sum += h[:, :15, 1].sum(axis=0)
        total += int(h[:, 15, 0].sum())
        vmax = max(vmax, int(h[:, 15, 1].max()))
        assert (h[:, :15, 0].sum(axis=1) == h[:, 15, 0]).all()  # per key: buckets add up to total_count_
        if first in (0, 7_000_000):  # digests of 2 x 10^6 keys: own totals / sums == the exact histogram of the same key
            if td_cap:  # (a quarter of them with the larger buffers: the host copy of the buffered values is 4 B x 1 920 per key)
                step, h = 250_000, h[:250_000]
            sums, cnts, mm = eng.export_tdigest(first, step)
            npend, pend = eng.export_tdigest_pending(first, step)
            assert (cnts.sum(axis=1) + npend == h[:, 15, 0]).all()
            psum = np.where(pend >= 0, pend, 0).astype(np.int64).sum(axis=1)
            assert (sums.sum(axis=1) + psum == h[:, :15, 1].sum(axis=1)).all()
            seen = h[:, 15, 0] > 0  # a key without events keeps the empty sentinels (INT32 / INT64 minimum)
            assert (mm[seen, 1] == h[seen, 15, 1]).all()  # digest max == max_val_seen_
            # sortedness: cluster means are non-decreasing in the cluster index (exact rational compare), min <= first mean, last <= max
            sel = np.arange(0, step, 997)
            for k in sel:
                nz = np.nonzero(cnts[k])[0]
                if len(nz) > 1:
                    s_, c_ = sums[k][nz].astype(object), cnts[k][nz].astype(object)
                    assert all(s_[i] * c_[i + 1] <= s_[i + 1] * c_[i] for i in range(len(nz) - 1))
                if len(nz):
                    assert mm[k, 0] * int(cnts[k][nz[0]]) <= int(sums[k][nz[0]]) and int(sums[k][nz[-1]]) <= mm[k, 1] * int(cnts[k][nz[-1]])
            dig_total += int(cnts.sum()) + int(npend.sum())
            dig_sum += int(sums.sum()) + int(psum.sum())
            step = 1_000_000
    assert total == 4 * n and vmax == max(max_a, max_b)
    assert (tot_cnt == 2 * gh_ab[:, 0]).all() and (tot_sum == 2 * gh_ab[:, 1]).all()
    assert dig_total > 0 and dig_sum > 0
    # ---- idempotence: queries (merged view of clusters + buffer) do not change any state
    g = int(wire.glob_id(1234, 567))
    s0, c0, m0 = eng.export_tdigest(eng.lookup(g), 1)
    q1 = eng.quantiles(g, [0.5, 0.95, 0.99])
    q2 = eng.quantiles(g, [0.5, 0.95, 0.99])
    s1, c1, m1 = eng.export_tdigest(eng.lookup(g), 1)
    assert q1 == q2 and (s0 == s1).all() and (c0 == c1).all() and (m0 == m1).all()
    assert q1[0] <= q1[1] <= q1[2]
    # window view after the last close is empty for every key; all-time view is unchanged by the close
    assert eng.export_hist(0, 5_000_000, 1000)[:, :15].sum() == 0
    slice_compare(False)
    # ---- the slice again after its keys have re-clustered inside the 10^7-key engine: 24 batches aimed at the first 25 hosts
    # (~42 values per key and batch: every key crosses the 896-value buffer once, i.e. 25 000 merges of steady-state size), two windows
    nb = 1 << 21  # (~84 values per key and batch: every key passes its buffer size with a wide margin)
    rounds = 24 if not td_cap else 24 * -(-td_cap // 896)  # (1 920-value buffers: 72 batches, every key re-clusters two or three times)
    for r in range(rounds):
        sg = eng.gen_resp_events(bufs[0].data_ptr(), nb, 0xC300 + r, 0, 25, sp)
        eng.handle_resp_events_dev(sg, bufs[0].data_ptr(), nb)
        eng.sync()
        oracle_feed(bufs[0], sg, nb)
        if r % 12 == 11 and r + 1 < rounds:
            eng.window_close()
            orc.window_clear(clear_hist=False)
    assert eng.counters()["td_merges"] >= 25 * sp
    slice_compare(True)
    g = int(wire.glob_id(3, 77))
    assert eng.quantiles(g, [0.25, 0.5, 0.99]) == [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(3 * sp + 77)), q) for q in (0.25, 0.5, 0.99)]
    eng.close()


# ---------------------------------------------------------------------------------------------------------------- C1
@TD_CAPS
@pytest.mark.parametrize("resp_path", [0, 1, 2, 3], ids=["auto-split", "general", "hostlocal-tiled", "hostlocal-split"])
def test_c1_single_host_replay_bit_exact(torch_mod, oracle, resp_path, td_cap):
    """ONE host, 100 services, a long replay (2^24 response events in one call, then 2^22 more): every key gets
    ~10^5 values per call (k_digest_huge with buffered values joining the merge), the single segment is far longer than any LDS image:
    the split form of the host-local pipeline (256 / 64 parts of 65536 events; what the time model picks), the general pipeline and the
    fused host-local pipeline with its tiled LDS image when forced.  Everything bit-exact vs the C oracle."""
    torch = torch_mod
    sp = 100
    eng = _engine(max_hosts=1, max_services=sp, max_batch_events=1 << 24, resp_path=resp_path, td_pend_cap=td_cap)
    orc = oracle.OracleEngine(sp, td_cap=td_cap)
    helpers.register_world(eng, orc, range(1), sp)
    for n, seed in ((1 << 24, 0xC1), (1 << 22, 0xC2), (5000, 0xC3)):
        ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        segs = eng.gen_resp_events(ev.data_ptr(), n, seed, 0, 1, sp)
        eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
        eng.sync()
        orc.resp_batch(ev.cpu().numpy().tobytes(), [0], [0])
    eng.sync()
    c = eng.counters()
    want = {0: (0, 1, 2), 1: (3, 0, 0), 2: (0, 3, 0), 3: (0, 1, 2)}[resp_path]  # (general, fused host-local, split); the 5000-event call is never split
    assert (c["resp_batches_general"], c["resp_batches_host_local"], c["resp_batches_host_split"]) == want
    helpers.assert_hist_equal(eng.export_hist(0, 0, sp), orc.hist(), sp)
    assert (eng.export_conn_bitmap(0, sp) == orc.bitmap()).all()
    gs, gc, gm = eng.export_tdigest(0, sp)
    os_, oc, om = orc.td_arrays()
    assert (gc == oc).all() and (gs == os_).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, sp)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    for s_idx in (0, 57, 99):
        g = int(wire.glob_id(0, s_idx))
        qs = [0.001, 0.5, 0.95, 0.999]
        assert eng.quantiles(g, qs) == [oracle.lib().gyo_tdb_quantile(C.byref(orc.td(s_idx)), q) for q in qs]
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all() and (eng.export_cms(0) == orc.cms()).all()
    eng.close()


# ---------------------------------------------------------------------------------------------------------------- C5
@TD_CAPS
def test_c5_zipf_heavy_hitters_bit_exact(torch_mod, oracle, td_cap):
    """10^5 services (50 hosts x 2000 listeners: close to the largest LDS sub-tables), Zipf(1.1) over a host's services: the head keys get
    10^5+ values per batch (k_digest_huge), the tail a handful (buffer appends).  Bit-exact vs the C oracle; CMS top-50 == exact top-50."""
    torch = torch_mod
    nh, sp, n = 50, 2000, 1 << 22
    nsvc = nh * sp
    eng = _engine(max_hosts=nh, max_services=nsvc, max_batch_events=n, resp_path=2, td_pend_cap=td_cap)  # 50 long segments: prefer host-local explicitly
    orc = oracle.OracleEngine(nsvc, td_cap=td_cap)
    helpers.register_world(eng, orc, range(nh), sp)
    ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    for rnd in range(3):
        segs = eng.gen_resp_events(ev.data_ptr(), n, 0xC5 + rnd, 0, nh, sp, 1100)
        eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
        eng.sync()
        orc.resp_batch(ev.cpu().numpy().tobytes(), [s.host_slot for s in segs], [s.first_event for s in segs])
    eng.sync()
    c = eng.counters()
    assert c["resp_batches_host_local"] == 3
    helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc.hist(), nsvc)
    assert (eng.export_conn_bitmap(0, nsvc) == orc.bitmap()).all()
    gs, gc, gm = eng.export_tdigest(0, nsvc)
    os_, oc, om = orc.td_arrays()
    assert (gc == oc).all() and (gs == os_).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, nsvc)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    hist = orc.hist()
    exact = hist[:, 15, 0].astype(np.int64)
    assert exact.max() > 1024 * 3  # the head really went through the huge-key kernel
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all() and (eng.export_cms(0) == orc.cms()).all()
    # heavy hitters: Count-Min estimate of every key >= exact, <= exact + eps*N (eps = e / 2^16), and the top-50 sets agree
    gids = np.concatenate([wire.glob_id(np.full(sp, h), np.arange(sp)) for h in range(nh)])
    top_exact = np.argsort(-exact, kind="stable")[:50]
    est_top = np.array([eng.cms(int(gids[k]), 0) for k in top_exact], dtype=np.int64)
    assert (est_top >= exact[top_exact]).all() and (est_top - exact[top_exact] <= np.e / 65536 * 3 * n).all()
    cms = eng.export_cms(0).astype(np.int64)
    def cms_cols(g):
        out = np.zeros(4, dtype=np.uint32)
        w = oracle.glob_id_words(int(g))
        oracle.lib().gyo_cms_cols(oracle.ptr(w, oracle.u32p), 2, oracle.ptr(out, oracle.u32p))
        return out
    cols = np.array([cms_cols(g) for g in gids[np.argsort(-exact)[:2000]]])
    est2000 = np.min([cms[r][cols[:, r]] for r in range(4)], axis=0)
    order2000 = np.argsort(-exact)[:2000]
    assert set(order2000[np.argsort(-est2000, kind="stable")[:50]].tolist()) == set(top_exact.tolist())
    eng.close()


# ---------------------------------------------------------------------------------------------------------------- C2
def test_c2_conn_and_listener_state_full_size(torch_mod, oracle):
    """1 000 hosts x 100 services; one window of 2^20 TCP_CONN_NOTIFY (20 % reconnects) + 10^5 LISTENER_STATE_NOTIFY records.
    HLL / both Count-Min tables bit-exact vs the C oracle; per-service counters, per-host summaries and cluster sums vs numpy."""
    torch = torch_mod
    rng = np.random.default_rng(2)
    nh, sp, n = 1000, 100, 1 << 20
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    mids = _register_bulk(eng, nh, sp)
    truth = {}
    rec = wire.synth_tcp_conns(rng, n, np.arange(nh), sp, dup_frac=0.2, v6_frac=0.05, truth=truth)
    raw = rec.tobytes()
    d_batch = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    d_off = torch.arange(0, n * 280, 280, dtype=torch.int32, device="cuda")
    from gyeeta_amd import capi
    eng.order()
    capi.check(eng.L.gys_ingest_tcp_conn_dev(eng.h, C.c_void_p(d_batch.data_ptr()), C.c_void_p(d_off.data_ptr()), n))
    hll = np.zeros(1 << 14, dtype=np.uint8)
    cms32 = np.zeros(4 * 65536, dtype=np.uint32)
    cms64 = np.zeros(4 * 65536, dtype=np.uint64)
    buf = np.frombuffer(raw, dtype=np.uint8)
    assert oracle.lib().gyo_tcp_conn_sketch_batch(buf.ctypes.data, n, buf.ctypes.data + len(buf), oracle.ptr(hll, oracle.u8p),
                                                   oracle.ptr(cms32, oracle.u32p), oracle.ptr(cms64, oracle.u64p)) == n
    # listener state: every host reports all its services once
    exp_summ = {}
    for h in range(nh):
        ls = wire.synth_listener_states(rng, h, np.arange(sp), bad_state_frac=0.01)
        eng.partha_listener_state(mids[h], ls.tobytes(), sp)
        ok = ls["curr_state"] <= 5
        st = np.bincount(ls["curr_state"][ok], minlength=6)[:6]
        exp_summ[h] = tuple(int(x) for x in st) + (int((ls["nqrys_5s"][ok] // 5).sum()), int(ls["nconns_active"][ok].sum()),
                                                     int(ls["curr_kbytes_inbound"][ok].sum()), int(ls["curr_kbytes_outbound"][ok].sum()),
                                                     int(ls["ser_errors"][ok].sum()), int(ok.sum()), int((ls["nqrys_5s"][ok] > 0).sum()))
        if h % 10 == 0:
            eng.handle_host_state(mids[h], ntasks=50, nlisten=sp)
    eng.window_close()
    assert (eng.export_hll() == hll).all()
    assert (eng.export_cms(0).ravel() == cms32).all()
    assert (eng.export_cms(1).ravel().astype(np.uint64) == cms64).all()
    # exact per-service counters vs the generator's CONNECTION table (slot = h * sp + s): every connection once -- whether it was
    # reported at its open and again at its close, by one half or by both, or as a loopback record -- with its closes and its bytes
    ctr = eng.export_svc_counters()
    slots = truth["conn_host"].astype(np.int64) * sp + truth["conn_svc"]
    assert (ctr[:, 0] == np.bincount(slots, minlength=nh * sp)).all() and len(slots) < n
    assert (ctr[:, 1] == np.bincount(slots[truth["conn_closed"]], minlength=nh * sp)).all()
    for col, f in ((2, "conn_bytes_sent"), (3, "conn_bytes_rcvd")):
        exp = np.zeros(nh * sp, dtype=np.uint64)
        np.add.at(exp, slots, truth[f])
        assert (ctr[:, col] == exp).all()
    # distinct flows vs the exact tuple set
    keys = np.concatenate([np.ascontiguousarray(rec[f]).view(np.uint8).reshape(n, 32) for f in ("nat_cli", "nat_ser")], axis=1)
    ndistinct = len(np.unique(keys, axis=0))
    assert abs(eng.distinct_flows() - ndistinct) / ndistinct < 0.03
    # LISTEN_SUMM_STATS of every host, and the cluster sums over the hosts that sent a host state
    tot_qps = {}
    for h in range(nh):
        assert eng.svcsumm(mids[h]).as_tuple() == exp_summ[h], h
        if h % 10 == 0:
            k = "cluster%d" % (h % 4)
            tot_qps[k] = tot_qps.get(k, 0) + exp_summ[h][6]
    for k, v in tot_qps.items():
        cs = eng.clusterstate(k)
        assert cs.total_qps == v and cs.nsvc == sp * sum(1 for h in range(0, nh, 10) if "cluster%d" % (h % 4) == k)
    eng.close()


# ---------------------------------------------------------------------------------------------------------------- C3, IPv6 stream
def test_c3_full_size_ipv6_slice_bench_config(torch_mod, oracle):
    """bench.py --ipv6 at its own size and gys_config (10 000 hosts x 1 000 services, td_pend_cap 1 920): the C3 traffic as 48-byte
    tcp_ipv6_resp_event_t events (2001:db8:: servers, fd00:: clients; bench.py's conversion of the generated IPv4 batch) through
    gys_ingest_resp_events_v6_dev.  A 50-host oracle slice (handle_ipv6_resp_event, common/gy_socket_stat.cc:1535-1551) is fed the same bytes:
    records, both CONN_BITMAP families, digests and buffered values bit for bit; counters and totals over all 10^7 keys."""
    torch = torch_mod
    nh, sp, n = 10_000, 1_000, 1 << 26
    nsvc = nh * sp
    eng = _engine(max_hosts=nh, max_services=nsvc, max_batch_events=n, td_pend_cap=BENCH_TD_PEND_CAP)
    _register_bulk(eng, nh, sp)
    SLICE = list(range(25)) + list(range(nh - 25, nh))
    orc = oracle.OracleEngine(len(SLICE) * sp, td_cap=BENCH_TD_PEND_CAP)
    for j, h in enumerate(SLICE):
        s = np.arange(sp)
        g, ns, pt = wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s)
        for i in range(sp):
            orc.register(j, int(g[i]), int(ns[i]), int(pt[i]))
    ev4 = torch.empty(n * 24, dtype=torch.uint8, device="cuda")

    def as_v6(nev):
        e4 = ev4[:nev * 24].view(torch.int32).view(-1, 6)
        e6 = torch.zeros((nev, 12), dtype=torch.int32, device="cuda")
        e6[:, 0] = 0xB80D0120 - (1 << 32)  # 20 01 0d b8
        e6[:, 3] = e4[:, 0]
        e6[:, 4] = 0x000000FD                # fd00::/8
        e6[:, 7] = e4[:, 1]
        e6[:, 8:12] = e4[:, 2:6]
        return e6.view(torch.uint8).view(-1)

    def feed(sg, nev, hosts_present):
        d6 = as_v6(nev)
        eng.handle_resp_events_v6_dev(sg, d6.data_ptr(), nev)
        eng.sync()
        for j, h in enumerate(SLICE):
            idx = next((k for k, x in enumerate(sg) if x.host_slot == h), None)
            if idx is None:
                continue
            lo = sg[idx].first_event
            hi = sg[idx + 1].first_event if idx + 1 < len(sg) else nev
            orc.resp_batch_v6(d6[lo * 48:hi * 48].cpu().numpy().tobytes(), [j], [0])
        del d6

    def slice_compare():
        for part, first in ((0, 0), (1, nsvc - 25 * sp)):
            k0, k1 = part * 25 * sp, (part + 1) * 25 * sp
            helpers.assert_hist_equal(eng.export_hist(1, first, 25 * sp), orc.hist()[k0:k1], 25 * sp)
            gb, ob = eng.export_conn_bitmap(first, 25 * sp), orc.bitmap()[k0:k1]
            assert (gb == ob).all() and not gb[:, :32].any()  # IPv6 events land in resp_bitmap_v6_ only
            gs, gc, gm = eng.export_tdigest(first, 25 * sp)
            os_, oc, om = orc.td_arrays()
            assert (gc == oc[k0:k1]).all() and (gs == os_[k0:k1]).all() and (gm == om[k0:k1]).all()
            gn, gp = eng.export_tdigest_pending(first, 25 * sp)
            on, op = orc.td_pending()
            assert (gn == on[k0:k1]).all() and (gp == op[k0:k1]).all()

    for b in range(2):
        sg = eng.gen_resp_events(ev4.data_ptr(), n, 0xC6 + b, 0, nh, sp)
        eng.sync()
        feed(sg, n, nh)
    slice_compare()
    c = eng.counters()
    assert c["resp_events"] == 2 * n and c["resp_dropped_range"] == 0 and c["resp_dropped_nolistener"] == 0 and c["resp_batches_general"] == 0
    eng.window_close()
    gh = eng.export_global_hist()
    assert gh.total_count == 2 * n and sum(gh.stats[i].count for i in range(15)) == 2 * n
    assert eng.export_cms(0).sum(axis=1).tolist() == [2 * n] * 4
    assert abs(eng.distinct_flows() - 2 * n) / (2 * n) < 0.04
    orc.window_clear(clear_hist=False)
    # the slice's keys past their 1 920-value buffers (merges inside the 10^7-key engine), IPv6 events only
    nb = 1 << 21
    for r in range(30):
        sg = eng.gen_resp_events(ev4.data_ptr(), nb, 0xC600 + r, 0, 25, sp)
        eng.sync()
        feed(sg, nb, 25)
    assert eng.counters()["td_merges"] >= 25 * sp
    slice_compare()
    eng.close()
