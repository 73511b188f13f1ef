"""GPU parity for the madhava entry points: partha_tcp_conn_info (TCP_CONN_NOTIFY -> distinct-flow HLL, CMS, per-service
counters), partha_listener_state (LISTENER_STATE_NOTIFY -> LISTEN_SUMM_STATS, top-N) and send_cluster_state (STATE_ONE sums)."""
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
        pytest.fail("no HIP device visible")
    return torch


def _engine(**kw):
    from gyeeta_amd.engine import SketchEngine
    return SketchEngine(**kw)


def _oracle_conn(oracle, batch, n, hll, cms32, cms64, known):
    """oracle decode of a variable-stride TCP_CONN_NOTIFY batch + sketch updates; returns per-glob_id exact counters"""
    L = oracle.lib()
    buf = np.frombuffer(batch, dtype=np.uint8)
    kw = np.zeros(n * 10, dtype=np.uint32)
    nw = np.zeros(n, dtype=np.uint32)
    gid = np.zeros(n, dtype=np.uint64)
    bs = np.zeros(n, dtype=np.uint64)
    br = np.zeros(n, dtype=np.uint64)
    fl = np.zeros(n, dtype=np.uint8)
    got = L.gyo_tcp_conn_decode(buf.ctypes.data, n, buf.ctypes.data + len(buf), oracle.ptr(kw, oracle.u32p), oracle.ptr(nw, oracle.u32p),
                                oracle.ptr(gid, oracle.u64p), oracle.ptr(bs, oracle.u64p), oracle.ptr(br, oracle.u64p), oracle.ptr(fl, oracle.u8p))
    assert got == n
    ctr = {}
    nlistener = 0
    for i in range(n):
        w = kw[i * 10:i * 10 + nw[i]]
        L.gyo_hll_add_words(oracle.ptr(hll, oracle.u8p), 14, oracle.ptr(w, oracle.u32p), int(nw[i]))
        # flag bits of gyo_tcp_conn_decode: 1 connect event, 2 accept event, 4 loopback, 8 pre-existing, 16 notified before.  A connection counts
        # once: on the record of its ACCEPTING half (or of neither kind: the walk treats it as the server's, gy_mconnhdlr.cc:9333) whose
        # notified_before_ is clear
        f = int(fl[i])
        if not ((f & 2) or not (f & 1)):
            continue
        nlistener += 1
        gw = oracle.glob_id_words(int(gid[i]))
        if not (f & 16):
            L.gyo_cms_add(oracle.ptr(cms32, oracle.u32p), oracle.ptr(gw, oracle.u32p), 2, 1)
        if int(bs[i]) + int(br[i]):
            L.gyo_cms64_add(oracle.ptr(cms64, oracle.u64p), oracle.ptr(gw, oracle.u32p), 2, int(bs[i]) + int(br[i]))
        if int(gid[i]) in known:
            c = ctr.setdefault(int(gid[i]), [0, 0, 0, 0])
            c[0] += 0 if (f & 16) else 1
            c[2] += int(bs[i])
            c[3] += int(br[i])
    return ctr, gid, nlistener


def test_tcp_conn_info_v4_v6_variable_stride(torch_mod, oracle):
    rng = np.random.default_rng(21)
    nh, sp = 4, 9
    eng = _engine(max_hosts=8, max_services=64, enable_tdigest=False)
    info, gids = helpers.register_world(eng, None, range(nh), sp)
    known = {int(g) for h in range(nh) for g in gids[h]}
    hll = np.zeros(1 << 14, dtype=np.uint8)
    cms32 = np.zeros(4 * 65536, dtype=np.uint32)
    cms64 = np.zeros(4 * 65536, dtype=np.uint64)
    exact = {}
    nclose = {}
    tuples = set()
    nlistener = nknown = 0
    tally = np.zeros(4, dtype=np.uint64)
    for rnd in range(3):
        for h in range(nh):
            n = int(rng.integers(1, 2049))  # MAX_NUM_CONNS = 2048 per message
            rec = wire.synth_tcp_conns(rng, n, list(range(nh)) + [77], sp, dup_frac=0.2, v6_frac=0.15)  # host 77: unregistered services
            tails = [bytes(rng.integers(32, 127, int(k), dtype=np.uint8).tolist()) for k in rng.integers(0, 257, n) * (rng.random(n) < 0.3)]
            batch = wire.pack_variable(rec, tails)
            eng.partha_tcp_conn_info(info[h][0], batch, n)
            ctr, gid, nl = _oracle_conn(oracle, batch, n, hll, cms32, cms64, known)
            nlistener += nl
            bb = np.frombuffer(batch, dtype=np.uint8)
            assert oracle.lib().gyo_tcp_conn_walk_tallies(bb.ctypes.data, n, bb.ctypes.data + len(bb), oracle.ptr(tally, oracle.u64p)) == n
            for g, c in ctr.items():
                e = exact.setdefault(g, [0, 0, 0, 0])
                for k in range(4):
                    e[k] += c[k]
            for i in range(n):
                lis = rec["is_tcp_accept_event"][i] or not rec["is_tcp_connect_event"][i]
                if lis and int(rec["ser_glob_id"][i]) in known:
                    nknown += 1
                    if rec["tusec_close"][i]:
                        nclose[int(rec["ser_glob_id"][i])] = nclose.get(int(rec["ser_glob_id"][i]), 0) + 1
                tuples.add((rec["nat_cli"][i].tobytes(), rec["nat_ser"][i].tobytes()))
    eng.window_close()
    assert (eng.export_hll() == hll).all()
    assert (eng.export_cms(0).ravel() == cms32).all()
    assert (eng.export_cms(1).ravel().astype(np.uint64) == cms64).all()
    ctrs = eng.export_svc_counters()
    for g, e in exact.items():
        s = eng.lookup(g)
        assert ctrs[s].tolist() == [e[0], nclose.get(g, 0), e[2], e[3]]
    c = eng.counters()
    assert c["conn_unknown_service"] > 0 and nlistener == nknown + c["conn_unknown_service"]
    # the walk's own tallies (nnew / nclosed / nclosed_no_not of gy_mconnhdlr.cc:9133-9137, :9327) and the connecting-half records
    assert [c["conn_new"], c["conn_closed"], c["conn_closed_no_notify"], c["conn_client_side"]] == tally.tolist()
    assert c["conn_events"] == c["conn_new"] + c["conn_closed"] and c["conn_client_side"] == c["conn_events"] - nlistener > 0
    # HLL estimate vs the exact distinct flow count (ground truth = exact set over the key bytes): p=14 -> ~0.8 % std error
    est = eng.distinct_flows()
    assert abs(est - len(tuples)) / len(tuples) < 0.05
    # malformed batch: truncated record is rejected, nothing ingested
    from gyeeta_amd import capi
    with pytest.raises(capi.GysError):
        eng.partha_tcp_conn_info(info[0][0], batch[:-8], n)
    eng.close()


def test_listener_state_summ_topn_cluster(torch_mod, oracle):
    rng = np.random.default_rng(33)
    L = oracle.lib()
    nh, sp = 6, 300
    eng = _engine(max_hosts=8, max_services=nh * sp, enable_tdigest=False, max_clusters=4)
    for c in ("cluster0", "cluster1", "cluster2"):
        eng.register_cluster(c)
    info, gids = helpers.register_world(eng, None, range(nh), sp)
    summ = {}
    recs = {}
    hstate = {}
    for h in range(nh):
        st = dict(ntasks_issue=int(rng.integers(0, 3)), ntasks=int(rng.integers(10, 500)), nlisten_issue=int(rng.integers(0, 2)),
                  nlisten=sp, cpu_issue=int(rng.integers(0, 2)), mem_issue=int(rng.integers(0, 2)))
        hstate[h] = st
        if h != 4:  # host 4 never reports a host state: send_cluster_state skips it (gy_mconnhdlr.cc:16068)
            eng.handle_host_state(info[h][0], **st)
        rec = wire.synth_listener_states(rng, h, np.arange(sp), delete_frac=0.01, bad_state_frac=0.01)
        rec["glob_id"][5] = 0xDEADBEEF12345  # unknown listener -> nmissed
        recs[h] = rec
        s = oracle.ListenSummStats()
        nerr = C.c_int(0)
        # the host's states arrive split over several messages with issue strings (variable stride); the window accumulates them
        for lo in range(0, sp, 128):
            part = rec[lo:lo + 128]
            tails = [b"issue!" * int(k) for k in rng.integers(0, 4, len(part))]
            batch = wire.pack_variable(part, tails)
            eng.partha_listener_state(info[h][0], batch, len(part))
            known_part = part[part["glob_id"] != 0xDEADBEEF12345]
            kb = wire.pack_variable(known_part, None)
            kbuf = np.frombuffer(kb, dtype=np.uint8)
            L.gyo_listener_state_rollup(kbuf.ctypes.data, len(known_part), kbuf.ctypes.data + len(kbuf), C.byref(s), C.byref(nerr))
        summ[h] = s
    eng.window_close()
    for h in range(nh):
        assert eng.svcsumm(info[h][0]).as_tuple() == summ[h].as_tuple()
    c = eng.counters()
    assert c["lstate_missed"] == nh and c["lstate_deleted"] > 0 and c["lstate_errors"] > 0
    # cluster state: CLUSTER_STATE_ONE::update_from_state over the hosts of each cluster that reported a host state
    for cl in range(3):
        exp = oracle.ClusterStateOne()
        for h in range(nh):
            if h % 3 == cl and h != 4:
                st = hstate[h]
                L.gyo_cluster_state_update(C.byref(exp), st["ntasks_issue"], st["ntasks"], st["nlisten_issue"], st["nlisten"],
                                           st["cpu_issue"], st["mem_issue"], C.byref(summ[h]))
        assert eng.clusterstate("cluster%d" % cl).as_tuple() == exp.as_tuple()
    # top-N per host: retained metric multiset == BOUNDED_PRIO_QUEUE result on the admitted records (gy_mconnhdlr.cc:11260-11304)
    for h in (0, 3):
        rec = recs[h]
        ok = (rec["glob_id"] != 0xDEADBEEF12345) & (rec["query_flags"] != wire.LISTEN_FLAG_DELETE) & (rec["curr_state"] <= 5)
        r = rec[ok]
        for kind, vals in ((1, r["nqrys_5s"][r["nqrys_5s"] >= 5]), (2, r["nconns_active"][r["nconns_active"] >= 1]),
                           (3, (r["curr_kbytes_inbound"].astype(np.int64) + r["curr_kbytes_outbound"])[(r["curr_kbytes_inbound"] + r["curr_kbytes_outbound"]) > 0])):
            v = np.ascontiguousarray(vals, dtype=np.uint64)
            out = np.zeros(10, dtype=np.uint64)
            k = L.gyo_topn_u64(oracle.ptr(v, oracle.u64p), len(v), 10, oracle.ptr(out, oracle.u64p))
            got = eng.topn(info[h][0], kind)
            assert [m for _, m, _ in got] == out[:k].tolist()
            for g, m, state in got:  # the entry carries the 88-byte record as ingested
                srec = np.frombuffer(state, dtype=wire.LISTENER_STATE_NOTIFY)[0]
                assert srec["glob_id"] == g
        issue = r[r["curr_state"] > 2]
        got = eng.topn(info[h][0], 0)
        exp = sorted(zip(issue["curr_state"].tolist(), issue["tasks_delay_usec"].tolist()), reverse=True)[:10]
        assert [(int(np.frombuffer(s, dtype=wire.LISTENER_STATE_NOTIFY)[0]["curr_state"]),
                 int(np.frombuffer(s, dtype=wire.LISTENER_STATE_NOTIFY)[0]["tasks_delay_usec"])) for _, _, s in got] == exp
    # a second window starts from zero
    eng.window_close()
    assert eng.svcsumm(info[0][0]).as_tuple() == (0,) * 13
    eng.close()


@pytest.mark.parametrize("kind", range(8))
def test_standalone_hist_all_kinds(torch_mod, oracle, kind):
    """rows a1-a4 for every reference hash class: keyed add, merge (add_histogram) and the per-key percentile scan"""
    torch = torch_mod
    from gyeeta_amd import capi
    rng = np.random.default_rng(40 + kind)
    eng = _engine(max_hosts=1, max_services=1, enable_tdigest=False)
    L = eng.L
    nk, n = 257, 200_000
    keys = rng.integers(0, nk, n).astype(np.uint32)
    vals = np.concatenate([rng.integers(-20, 400, n // 2), (rng.lognormal(5, 3, n - n // 2)).clip(0, 2**31 - 1)]).astype(np.int32)
    vals[:8] = [-1, 0, 1, 2**31 - 1, -2**31, 15000, 15001, 5000001]
    dk, dv = torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(vals).cuda()
    h1 = torch.zeros(nk * 256, dtype=torch.uint8, device="cuda")
    h2 = torch.zeros(nk * 256, dtype=torch.uint8, device="cuda")
    half = n // 2
    eng.order()  # the torch fills / copies above precede the engine's kernels
    for hh, lo, hi in ((h1, 0, half), (h2, half, n)):
        capi.check(L.gys_hist_init_dev(eng.h, kind, hh.data_ptr(), nk))
        capi.check(L.gys_hist_add_dev(eng.h, kind, hh.data_ptr(), nk, dk[lo:hi].data_ptr(), dv[lo:hi].data_ptr(), hi - lo))
    capi.check(L.gys_hist_merge_dev(eng.h, h1.data_ptr(), h2.data_ptr(), nk))
    pcts = np.array([1, 25, 50, 75, 95, 99, 99.99, 100], dtype=np.float32)
    out = torch.zeros(nk * len(pcts), dtype=torch.int64, device="cuda")
    capi.check(L.gys_hist_percentiles_dev(eng.h, kind, h1.data_ptr(), nk, pcts.ctypes.data_as(capi.f32p), len(pcts), out.data_ptr()))
    eng.sync()
    got = h1.cpu().numpy().view(np.int64).reshape(nk, 16, 2)
    stats, total, maxv = oracle.keyed_hist(kind, nk, keys, vals)
    assert (got[:, :15, :] == stats[:, :15, :]).all()
    assert (got[:, 15, 0] == total.astype(np.int64)).all() and (got[:, 15, 1] == maxv).all()
    gp = out.cpu().numpy().reshape(nk, len(pcts))
    for k in range(0, nk, 17):
        ov, _, _, _ = oracle.hist_percentiles(kind, stats[k], total[k], [float(p) for p in pcts])
        assert gp[k].tolist() == ov
    eng.close()


def test_active_conn_stats_pair_countmin_and_listener_sums(oracle):
    """comm::ACTIVE_CONN_STATS roll-up (MCONN_HANDLER::handle_partha_active_conns): the Count-Min pair keyed by
    (listener, client task group) bit-exact vs the oracle's restatement, exact per-listener sums vs numpy, remote-listener rows only
    counted; estimates never below the exact per-pair totals (Count-Min over-estimates)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(41)
    nh, sp = 4, 12
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    gids = {}
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "c")
        s = np.arange(sp)
        gids[h] = wire.glob_id(np.full(sp, h), s)
        eng.register_listeners_np(mid, gids[h], wire.listener_netns(h, s), wire.listener_port(s))
    L = oracle.lib()
    for w in range(2):
        p32 = np.zeros((4, 65536), dtype=np.uint32)
        p64 = np.zeros((4, 65536), dtype=np.uint64)
        r32 = np.zeros((4, 65536), dtype=np.uint32)  # the rows whose listener lives on another madhava (is_remote_listen_ -> remoteconntbl): tables of their own
        r64 = np.zeros((4, 65536), dtype=np.uint64)
        tot = np.zeros(2, dtype=np.uint64)
        allrec = []
        for h in range(nh):
            for msg in range(3):
                n = int(rng.integers(1, 2049)) if msg else 2048  # MAX_NUM_CONNS rows per message
                rec = wire.synth_active_conns(rng, n, h, sp)
                raw = rec.tobytes()
                eng.handle_partha_active_conns(wire.machine_id(h), raw, n)
                o2 = np.zeros(2, dtype=np.uint64)
                buf = np.frombuffer(raw, dtype=np.uint8)
                L.gyo_active_conn_sketch_batch2(oracle.ptr(buf, oracle.u8p), n, oracle.ptr(p32, oracle.u32p), oracle.ptr(p64, oracle.u64p),
                                                oracle.ptr(r32, oracle.u32p), oracle.ptr(r64, oracle.u64p), oracle.ptr(o2, oracle.u64p))
                tot += o2
                allrec.append(rec)
        eng.window_close()
        assert (eng.export_pair_cms(0) == p32).all()
        assert (eng.export_pair_cms(1).view(np.uint64) == p64).all()
        assert r32.any() and (eng.export_pair_cms(6) == r32).all() and (eng.export_pair_cms(7).view(np.uint64) == r64).all()
        # a partha reports every 15 s, a window is 5 s: two windows without ACTIVE_CONN_STATS rows leave the last report readable
        eng.window_close()
        eng.window_close()
        assert (eng.export_pair_cms(0) == p32).all() and (eng.export_pair_cms(1).view(np.uint64) == p64).all()
        assert (eng.export_pair_cms(6) == r32).all() and (eng.export_pair_cms(7).view(np.uint64) == r64).all()
        rec = np.concatenate(allrec)
        local = (rec["flags"] & wire.ACTIVE_FLAG_REMOTE_LISTEN) == 0
        c = eng.counters()
        assert c["actconn_records"] == int(tot[0]) * 1 + (0 if w == 0 else prev_local) and c["actconn_remote_listen"] == int(tot[1]) + (0 if w == 0 else prev_remote)
        prev_local, prev_remote = c["actconn_records"], c["actconn_remote_listen"]
        # a few pairs: estimate >= exact total of the window, and == the oracle table's own estimate
        lr = rec[local]
        for k in rng.integers(0, len(lr), 6):
            g, t = int(lr["listener_glob_id"][k]), int(lr["cli_aggr_task_id"][k])
            sel = (lr["listener_glob_id"] == g) & (lr["cli_aggr_task_id"] == t)
            assert eng.pair_cms(g, t, 0) >= int(lr["active_conns"][sel].sum())
            assert eng.pair_cms(g, t, 1) >= int(lr["bytes_sent"][sel].sum() + lr["bytes_received"][sel].sum())
        rr = rec[~local]
        for k in rng.integers(0, len(rr), 4):  # remote-listener pairs: never below what was reported
            g, t = int(rr["listener_glob_id"][k]), int(rr["cli_aggr_task_id"][k])
            sel = (rr["listener_glob_id"] == g) & (rr["cli_aggr_task_id"] == t)
            assert eng.pair_cms(g, t, 6) >= int(rr["active_conns"][sel].sum())
            assert eng.pair_cms(g, t, 7) >= int(rr["bytes_sent"][sel].sum() + rr["bytes_received"][sel].sum())
        if w == 0:
            first = rec
    # exact cumulative per-listener sums over both windows
    rec = np.concatenate([first, rec])
    local = rec[(rec["flags"] & wire.ACTIVE_FLAG_REMOTE_LISTEN) == 0]
    got = eng.export_active_conn_counters()
    nunk = 0
    for h in range(nh):
        for s_ in range(sp):
            sel = local["listener_glob_id"] == gids[h][s_]
            exp = [int(sel.sum()), int(local["bytes_sent"][sel].sum()), int(local["bytes_received"][sel].sum()), int(local["active_conns"][sel].sum())]
            assert got[h * sp + s_].tolist() == exp
    known = np.isin(local["listener_glob_id"], np.concatenate(list(gids.values())))
    assert eng.counters()["actconn_unknown_listener"] == int((~known).sum()) > 0
    eng.close()


def test_active_conn_stats_staggered_reports_stay_visible_for_three_windows(oracle):
    """Parthas report ACTIVE_CONN_STATS every 15 s on phases of their own and a window is 5 s (server/gy_mconnhdlr.cc:7714): a window carries
    only the hosts that reported in it.  The queries read the per-cell MAXIMUM of the last three windows' tables: every host that reported in
    the last 15 s is visible with at least its reported value (Count-Min never under-estimates), a host that reported twice inside the three
    windows is not counted twice, and a host silent for three windows ages out.  Expected tables: numpy max over the oracle's per-window tables."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(47)
    nh, sp = 3, 8
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "c")
        s = np.arange(sp)
        eng.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s))
    L = oracle.lib()
    # which hosts report in which window: host h on phase h of the 15-s cycle, host 0 once more out of turn (window 4), then silence
    plan = [[0], [1], [2], [0], [0, 1], [2], [], [], [], [], [1]]
    tabs32, tabs64, recs = [], [], []
    for w, hosts in enumerate(plan):
        p32 = np.zeros((4, 65536), dtype=np.uint32)
        p64 = np.zeros((4, 65536), dtype=np.uint64)
        wrec = {}
        for h in hosts:
            n = int(rng.integers(200, 900))
            rec = wire.synth_active_conns(rng, n, h, sp)
            raw = rec.tobytes()
            eng.handle_partha_active_conns(wire.machine_id(h), raw, n)
            o2 = np.zeros(2, dtype=np.uint64)
            buf = np.frombuffer(raw, dtype=np.uint8)
            L.gyo_active_conn_sketch_batch(oracle.ptr(buf, oracle.u8p), n, oracle.ptr(p32, oracle.u32p), oracle.ptr(p64, oracle.u64p), oracle.ptr(o2, oracle.u64p))
            wrec[h] = rec[(rec["flags"] & wire.ACTIVE_FLAG_REMOTE_LISTEN) == 0]
        tabs32.append(p32)
        tabs64.append(p64)
        recs.append(wrec)
        eng.window_close()
        lo = max(0, w - 2)
        want32 = np.maximum.reduce(tabs32[lo:w + 1])
        want64 = np.maximum.reduce(tabs64[lo:w + 1])
        assert (eng.export_pair_cms(0) == want32).all(), w
        assert (eng.export_pair_cms(1).view(np.uint64) == want64).all(), w
        # every host that reported within the last three windows: its LATEST report reads back at least its exact per-pair totals
        latest = {}
        for ww in range(lo, w + 1):
            latest.update(recs[ww])
        for h, lr in latest.items():
            for k in rng.integers(0, len(lr), 4):
                g, t = int(lr["listener_glob_id"][k]), int(lr["cli_aggr_task_id"][k])
                sel = (lr["listener_glob_id"] == g) & (lr["cli_aggr_task_id"] == t)
                assert eng.pair_cms(g, t, 0) >= int(lr["active_conns"][sel].sum()), (w, h)
        if w == 9:  # four windows of silence: everything has aged out
            assert not eng.export_pair_cms(0).any() and not eng.export_pair_cms(1).any()
    eng.close()


def test_tcp_conn_pair_countmin_opt_in(oracle):
    """gys_config.conn_pair_cms (connlistenmap_ / connclientmap_ roll-up): TCP_CONN_NOTIFY records also feed the
    (listener, client task group) Count-Min pair, bit-exact vs the oracle; off by default (tables stay zero)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(43)
    nh, sp, n = 3, 10, 1500
    engs = {flag: _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False, conn_pair_cms=flag) for flag in (True, False)}
    for e in engs.values():
        for h in range(nh):
            mid = wire.machine_id(h)
            e.register_host(mid, "c")
            s = np.arange(sp)
            e.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s))
    p32 = np.zeros((4, 65536), dtype=np.uint32)
    p64 = np.zeros((4, 65536), dtype=np.uint64)
    c32 = np.zeros((4, 65536), dtype=np.uint32)
    c64 = np.zeros((4, 65536), dtype=np.uint64)
    L = oracle.lib()
    nlis = ncli = 0
    for h in range(nh):
        rec = wire.synth_tcp_conns(rng, n, [h], sp, dup_frac=0.3)
        rec["cli_task_aggr_id"] = wire.splitmix64(rng.integers(0, 25, n).astype(np.uint64) + np.uint64(h << 8))
        # what connlistenmap_ / connclientmap_ take (gy_mconnhdlr.cc:9182, :9226, :9290): closed records with bytes, by half
        cb = (rec["tusec_close"] != 0) & (rec["bytes_sent"] + rec["bytes_rcvd"] > 0)
        nlis += int((cb & (rec["is_tcp_accept_event"] != 0) & (rec["ser_glob_id"] != 0)).sum())
        ncli += int((cb & (rec["is_tcp_accept_event"] == 0) & (rec["is_tcp_connect_event"] != 0)).sum())
        tails = [bytes(rng.integers(32, 127, int(k), dtype=np.uint8).tolist()) for k in rng.integers(0, 40, n) * (rng.random(n) < 0.3)]
        payload = wire.pack_variable(rec, tails)
        for e in engs.values():
            e.partha_tcp_conn_info(wire.machine_id(h), payload, n)
        buf = np.frombuffer(payload, dtype=np.uint8)
        got = L.gyo_tcp_conn_pair_batch(oracle.ptr(buf, oracle.u8p), n, C.cast(buf.ctypes.data + len(buf), oracle.u8p), oracle.ptr(p32, oracle.u32p),
                                        oracle.ptr(p64, oracle.u64p), oracle.ptr(c32, oracle.u32p), oracle.ptr(c64, oracle.u64p))
        assert got == n
    for e in engs.values():
        e.window_close()
    assert (engs[True].export_pair_cms(2) == p32).all() and (engs[True].export_pair_cms(3).view(np.uint64) == p64).all()
    assert (engs[True].export_pair_cms(4) == c32).all() and (engs[True].export_pair_cms(5).view(np.uint64) == c64).all()
    # every row counts every closed connection once, on the side that reported it
    assert p32.sum(axis=1).tolist() == [nlis] * 4 and c32.sum(axis=1).tolist() == [ncli] * 4 and nlis > 0 and ncli > 0
    # the ACTIVE_CONN_STATS tables are a different roll-up with cells of their own: connection notifications never reach them
    assert engs[True].export_pair_cms(0).sum() == 0 and engs[True].export_pair_cms(1).sum() == 0
    assert engs[False].export_pair_cms(0).sum() == 0 and engs[False].export_pair_cms(1).sum() == 0
    with pytest.raises(Exception):
        engs[False].export_pair_cms(2)  # option off: GYS_ERR_STATE
    # the default registers are unaffected by the option
    assert (engs[True].export_cms(0) == engs[False].export_cms(0)).all() and (engs[True].export_hll() == engs[False].export_hll()).all()
    for e in engs.values():
        e.close()
