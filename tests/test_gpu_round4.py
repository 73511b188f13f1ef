"""GPU parity, round 4: the filtered multi-host listener-state query (QUERY_OPTIONS on the live table: criteria groups, sort, maxrecs, host
subset, AGGR_OPER_E reductions) against numpy on the records the test fed; the checker restates SvcStateFields::get_num_field
(server/gy_mfields.h:1402-1440) and CRITERIA_SET::match_criteria (common/gy_query_criteria.h:1535-1605, :1806-1900) independently of the
kernel."""
import json

import numpy as np
import pytest

from gyeeta_amd import capi, wire
from tests import helpers

pytestmark = pytest.mark.gpu


def _engine(**kw):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    from gyeeta_amd.engine import SketchEngine
    return SketchEngine(**kw)


def col_values(rec, col):
    """the column as the reference compares it: an `int` built from the wire field (unsigned arithmetic first), int16_t for `issue`"""
    nq = rec["nqrys_5s"].astype(np.uint32)
    u = {
        "qps5s": nq // 5, "nqry5s": nq, "resp5s": rec["total_resp_5sec"].astype(np.uint32) // np.maximum(nq, 1),
        "p95resp5s": rec["p95_5s_resp_ms"], "p95resp5m": rec["p95_5min_resp_ms"], "nconns": rec["nconns"], "nactive": rec["nconns_active"],
        "nprocs": rec["ntasks"], "kbin15s": rec["curr_kbytes_inbound"], "kbout15s": rec["curr_kbytes_outbound"], "sererr": rec["ser_errors"],
        "clierr": rec["cli_errors"], "delayus": rec["tasks_delay_usec"], "cpudelus": rec["tasks_cpudelay_usec"], "iodelus": rec["tasks_blkiodelay_usec"],
        "usercpu": rec["tasks_user_cpu"], "syscpu": rec["tasks_sys_cpu"], "rssmb": rec["tasks_rss_mb"], "nissue": rec["ntasks_issue"],
        "state": rec["curr_state"], "issue": rec["curr_issue"], "ishttp": (rec["is_http_svc"] != 0),
    }
    if col == "vmdelus":
        with np.errstate(over="ignore"):
            v = rec["tasks_delay_usec"].astype(np.uint32) - rec["tasks_cpudelay_usec"].astype(np.uint32) - rec["tasks_blkiodelay_usec"].astype(np.uint32)
        return v.view(np.int32).astype(np.int64)
    return u[col].astype(np.uint32).view(np.int32).astype(np.int64) if col not in ("issue", "ishttp") else u[col].astype(np.int64)


def match(rec, terms, group_oper=(), top_oper="and"):
    """CRITERIA_SET::match_criteria for criteria of one subsystem: per group all / any of its terms, groups combined by top_oper"""
    if not terms:
        return np.ones(len(rec), dtype=bool)
    groups = {}
    for t in terms:
        col, comp, val = t[0], t[1], t[2]
        g = t[3] if len(t) > 3 else 0
        v = col_values(rec, col)
        conv = (lambda x: int(np.int16(np.int64(x) & 0xFFFF)) if False else int(x))
        if comp in ("in", "notin"):
            m = np.isin(v, [conv(x) for x in val])
            m = m if comp == "in" else ~m
        elif comp == "bit2":
            m = (v & 3) == 3
        elif comp == "bit3":
            m = (v & 7) == 7
        else:
            c = conv(val)
            m = {"=": v == c, "!=": v != c, "<": v < c, "<=": v <= c, ">": v > c, ">=": v >= c}[comp]
        groups.setdefault(g, []).append(m)
    res = []
    for g, ms in sorted(groups.items()):
        o = group_oper[g] if g < len(group_oper) else "and"
        res.append(np.logical_or.reduce(ms) if o == "or" else np.logical_and.reduce(ms))
    return np.logical_or.reduce(res) if top_oper == "or" else np.logical_and.reduce(res)


def test_svcstate_filter_sort_topk_and_aggregation_equal_numpy():
    rng = np.random.default_rng(404)
    nh, sp = 40, 50
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False, max_clusters=4)
    for cname in ("cl0", "cl1", "cl2"):
        eng.register_cluster(cname)
    mids = [wire.machine_id(h) for h in range(nh)]
    s_ = np.arange(sp)
    for h in range(nh):
        eng.register_host(mids[h], "cl%d" % (h % 3))
        eng.set_host_name(mids[h], "host%03d" % h)
        eng.register_listeners_np(mids[h], wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))

    def states(h):
        r = wire.synth_listener_states(rng, h, s_)
        n = len(r)
        # widen the value ranges the synthetic generator leaves narrow, incl. values whose `int` view is negative and delay fields whose
        # unsigned difference wraps
        r["cli_errors"] = rng.integers(0, 9, n)
        r["tasks_cpudelay_usec"] = rng.integers(0, 150000, n)
        r["tasks_blkiodelay_usec"] = rng.integers(0, 50000, n)
        r["tasks_user_cpu"] = rng.integers(0, 400, n)
        r["tasks_sys_cpu"] = rng.integers(0, 100, n)
        r["tasks_rss_mb"] = rng.integers(1, 60000, n)
        r["ntasks_issue"] = rng.integers(0, 4, n)
        r["curr_issue"] = rng.integers(0, 20, n)
        r["is_http_svc"] = rng.integers(0, 2, n)
        big = rng.random(n) < 0.03
        r["curr_kbytes_inbound"] = np.where(big, 0xF0000000 + rng.integers(0, 1000, n), r["curr_kbytes_inbound"])
        return r

    latest = {}
    for h in range(nh):  # window A: every host
        r = states(h)
        eng.partha_listener_state(mids[h], r.tobytes(), sp)
        latest[h] = r
    eng.window_close()
    for h in range(30):  # window B: hosts 0..29 report again -- the others' states will be two windows old
        r = states(h)
        r["query_flags"][7] = wire.LISTEN_FLAG_DELETE  # ... and every host deletes one listener
        eng.partha_listener_state(mids[h], r.tobytes(), sp)
        latest[h] = r
    eng.window_close()
    for h in range(5):  # the open window: hosts 0..4 already reported
        r = states(h)
        eng.partha_listener_state(mids[h], r.tobytes(), sp)
        latest[h] = r
    eng.sync()
    # what is current: hosts 0..29, minus the listeners deleted in window B that did not report again
    rec = np.concatenate([latest[h] for h in range(30)])
    slot = np.concatenate([np.arange(sp) + h * sp for h in range(30)])
    host = np.repeat(np.arange(30), sp)
    alive = np.ones(len(rec), dtype=bool)
    for h in range(5, 30):
        alive[h * sp + 7] = False
    rec, slot, host = rec[alive], slot[alive], host[alive]

    q50 = int(np.median(col_values(rec, "qps5s")))
    cases = [
        dict(terms=None),
        dict(terms=[("qps5s", ">", q50), ("p95resp5s", ">=", 60)]),
        dict(terms=[("state", "=", 3), ("sererr", ">", 2), ("nissue", "!=", 0)], group_oper=["or"]),
        dict(terms=[("nactive", "in", [0, 3, 7, 11]), ("kbin15s", "<", 0)], group_oper=["or"]),      # kbin15s < 0: the u32 values above 2^31 as `int`
        dict(terms=[("issue", "notin", [1, 2, 3]), ("vmdelus", "<", 0), ("ishttp", "=", 1)]),
        dict(terms=[("nconns", "bit2", 0), ("nprocs", "bit3", 0, 1), ("rssmb", ">", 50000, 1)], group_oper=["and", "or"], top_oper="or"),
        dict(terms=[("qps5s", ">", q50, 0), ("resp5s", "<=", 20, 0), ("usercpu", ">", 300, 1), ("syscpu", ">", 80, 1), ("delayus", ">=", 50000, 2)],
             group_oper=["and", "or", "and"], top_oper="and"),
        dict(terms=[("qps5s", ">=", 0)], machine_ids=[mids[3], mids[17], mids[35], wire.machine_id(999)]),  # host 35: stale, 999: unknown
        # the query names its listeners (svcid = / in: the reference's direct-lookup path) -- some current, one deleted, one stale, one unknown
        dict(terms=[("nconns", ">=", 0)], svcids=[int(wire.glob_id(h_, s__)) for h_, s__ in ((0, 3), (2, 7), (9, 7), (9, 8), (29, 49), (33, 1))] + [12345]),
        dict(terms=[("qps5s", ">", q50)], clusters=["cl1", "nosuchcluster"]),
        dict(terms=None, clusters=["cl0", "cl2"], machine_ids=[mids[0], mids[1], mids[2], mids[3]]),          # hosts 0, 2, 3 (cl0, cl2, cl0)
    ]
    for ci, case in enumerate(cases):
        m = match(rec, case.get("terms"), case.get("group_oper", ()), case.get("top_oper", "and"))
        if case.get("machine_ids"):
            m &= np.isin(host, [h_ for h_ in range(30) if any(mids[h_] == x for x in case["machine_ids"])])
        if case.get("clusters"):
            m &= np.isin(host % 3, [int(c_[2:]) for c_ in case["clusters"] if c_.startswith("cl")])
        if case.get("svcids"):
            m &= np.isin(rec["glob_id"], np.array(case["svcids"], dtype=np.uint64))
        # AOPER_PERCENTILE: the discrete percentile of a column over the matching records (percentile_disc: the ceil(p N)-th smallest)
        pcts = [0.01, 0.25, 0.5, 0.95, 0.99, 1.0, 1e-9]
        for pcol in ("qps5s", "kbin15s", "vmdelus", "state"):
            got, nm_p = eng.svcstate_percentiles(pcol, pcts, case.get("terms"), case.get("group_oper", ()), case.get("top_oper", "and"),
                                                  case.get("machine_ids"), case.get("svcids"), case.get("clusters"))
            assert nm_p == int(m.sum()), (ci, pcol)
            if nm_p:
                sv = np.sort(col_values(rec[m], pcol).astype(np.int64))
                want_p = [int(sv[min(max(int(np.ceil(p_ * nm_p)), 1), nm_p) - 1]) for p_ in pcts]
                assert got.tolist() == want_p, (ci, pcol, got.tolist(), want_p)
            else:
                assert not got.any()
        for sort_col, desc in ((None, True), ("qps5s", True), ("p95resp5s", False), ("kbin15s", True), ("vmdelus", False)):
            if sort_col is None:
                order = np.argsort(slot[m], kind="stable")
            else:
                v = col_values(rec[m], sort_col)
                order = np.lexsort((slot[m], -v if desc else v))
            want_slots = slot[m][order]
            want_recs = rec[m][order]
            for maxrecs in (len(rec) + 5, 17, 1):
                gs, gh, gr, nm = eng.svcstate_scan(case.get("terms"), case.get("group_oper", ()), case.get("top_oper", "and"), sort_col, desc, maxrecs,
                                                   case.get("machine_ids"), case.get("svcids"), case.get("clusters"))
                assert nm == int(m.sum()), (ci, sort_col, maxrecs, nm, int(m.sum()))
                k = min(maxrecs, len(want_slots))
                assert gs.tolist() == want_slots[:k].tolist(), (ci, sort_col, desc, maxrecs)
                assert (gh == gs // sp).all()
                assert gr.tobytes() == want_recs[:k].tobytes(), (ci, sort_col, maxrecs)
        # the aggregation operators over the same matching set: global, per host, per cluster
        cols = ["qps5s", "resp5s", "kbin15s", "vmdelus", "ishttp", "nactive"]
        for group_by, gkey in ((0, np.zeros(len(rec), dtype=np.int64)), (1, host), (2, host % 3)):
            got = eng.svcstate_aggr(cols, group_by, case.get("terms"), case.get("group_oper", ()), case.get("top_oper", "and"), case.get("machine_ids"),
                                    svcids=case.get("svcids"), clusters=case.get("clusters"))
            want = []
            for g in np.unique(gkey[m]):
                sel = m & (gkey == g)
                want.append((int(g), int(sel.sum()), {c: (int(col_values(rec[sel], c).sum()), int(col_values(rec[sel], c).min()), int(col_values(rec[sel], c).max()))
                                                    for c in cols}))
            assert got == want, (ci, group_by)
    # operator values from a row (AGGR_OPER_E)
    import ctypes as C
    f, keep = eng._svc_filter(None)
    ca = (C.c_uint8 * 2)(capi.SVC_COLS.index("qps5s"), capi.SVC_COLS.index("ishttp"))
    row = (capi.SvcAggrRow * 1)()
    n = C.c_uint32()
    capi.check(eng.L.gys_query_svcstate_aggr(eng.h, C.byref(f), 0, ca, 2, row, 1, C.byref(n)))
    q = col_values(rec, "qps5s")
    out = C.c_double()
    for oper, want in (("sum", float(q.sum())), ("avg", float(q.sum()) / len(q)), ("max", float(q.max())), ("min", float(q.min())), ("count", float(len(q)))):
        capi.check(eng.L.gys_svc_aggr_value(row, 0, capi.AOPER[oper], C.byref(out)))
        assert out.value == want, oper
    capi.check(eng.L.gys_svc_aggr_value(row, 1, capi.AOPER["bool_or"], C.byref(out)))
    assert out.value == 1.0
    capi.check(eng.L.gys_svc_aggr_value(row, 1, capi.AOPER["bool_and"], C.byref(out)))
    assert out.value == 0.0
    # the multi-host JSON: envelope, the four host columns first, then the reference's svcstate columns; values of the top rows
    from tests import test_gpu_json as tj
    js = json.loads(eng.json_svcstate_multihost([("qps5s", ">", q50)], sort_col="qps5s", sort_desc=True, maxrecs=25, madid="ab" * 8, timestr="T"))
    assert list(js.keys()) == ["madid", "svcstate"] and js["madid"] == "ab" * 8 and len(js["svcstate"]) == 25
    m = match(rec, [("qps5s", ">", q50)])
    v = col_values(rec[m], "qps5s")
    order = np.lexsort((slot[m], -v))
    for i, row_ in enumerate(js["svcstate"]):
        assert list(row_.keys()) == ["parid", "host", "madid", "cluster"] + tj.SVCSTATE_COLS
        s0 = int(slot[m][order][i])
        r0 = rec[m][order][i]
        assert row_["svcid"] == "%016x" % int(r0["glob_id"]) and row_["qps5s"] == int(r0["nqrys_5s"]) // 5 and row_["host"] == "host%03d" % (s0 // sp)
        assert row_["parid"] == "%016x%016x" % (int.from_bytes(mids[s0 // sp][:8], "little"), int.from_bytes(mids[s0 // sp][8:], "little"))
        assert row_["cluster"] == "cl%d" % ((s0 // sp) % 3)
        with np.errstate(over="ignore"):
            vm = int((np.uint32(r0["tasks_delay_usec"]) - np.uint32(r0["tasks_cpudelay_usec"]) - np.uint32(r0["tasks_blkiodelay_usec"])))
        assert row_["vmdelus"] == vm  # the unsigned 32-bit difference, as the reference prints it (server/gy_mfields.h:1648-1654)
    # bad arguments
    with pytest.raises(capi.GysError):
        eng.svcstate_scan([("qps5s", ">", 1, 9)])
    eng.close()


def test_svcsumm_multihost_filter_equals_numpy():
    """web_curr_listener_summ, multi-host form: every host that reported in the last window, filtered on the LISTEN_SUMM_STATS columns,
    sorted, capped; envelope and column order of the reference (host columns first, then json_db_svcsumm_arr)"""
    from tests import test_gpu_json as tj
    rng = np.random.default_rng(77)
    nh, sp = 24, 30
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False, max_clusters=4)
    mids = [wire.machine_id(h) for h in range(nh)]
    s_ = np.arange(sp)
    for h in range(nh):
        eng.register_host(mids[h], "cl%d" % (h % 3))
        eng.set_host_name(mids[h], "host%03d" % h)
        eng.register_listeners_np(mids[h], wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))
    want = {}
    for h in range(nh):
        if h % 7 == 6:
            continue  # these hosts send nothing in the window: not listed
        ls = wire.synth_listener_states(rng, h, s_, bad_state_frac=0.05)
        eng.partha_listener_state(mids[h], ls.tobytes(), sp)
        ok = ls["curr_state"] <= 5
        st = np.bincount(ls["curr_state"][ok], minlength=6)[:6]
        want[h] = [int(x) for x in st] + [int((ls["nqrys_5s"][ok] // 5).sum()), int(ls["nconns_active"][ok].sum()), int(ls["curr_kbytes_inbound"][ok].sum()),
                                          int(ls["curr_kbytes_outbound"][ok].sum()), int(ls["ser_errors"][ok].sum()), int(ok.sum()), int((ls["nqrys_5s"][ok] > 0).sum())]
    eng.window_close()
    cols = capi.SUMM_COLS
    rows = np.array([want[h] for h in sorted(want)], dtype=np.int64)
    hosts = np.array(sorted(want))
    medq = int(np.median(rows[:, cols.index("totqps")]))
    cases = [
        dict(),
        dict(terms=[("totqps", ">", medq)]),
        dict(terms=[("nbad", ">", 2), ("nsevere", ">=", 2)], group_oper=["or"], sort_col="totaconn", sort_desc=True),
        dict(terms=[("totsererr", "<", 60, 0), ("nactive", ">=", 20, 1), ("ndown", "=", 0, 1)], group_oper=["and", "and"], top_oper="or", sort_col="totkbin", sort_desc=False, maxrecs=5),
        dict(terms=[("nsvc", ">=", 1)], clusters=["cl2"], sort_col="totqps"),
        dict(machine_ids=[mids[1], mids[6], mids[9]]),  # host 6 sent nothing
    ]
    for ci, case in enumerate(cases):
        m = np.ones(len(rows), dtype=bool)
        if case.get("terms"):
            groups = {}
            for t in case["terms"]:
                v = rows[:, cols.index(t[0])]
                mm = {"=": v == t[2], "!=": v != t[2], "<": v < t[2], "<=": v <= t[2], ">": v > t[2], ">=": v >= t[2]}[t[1]]
                groups.setdefault(t[3] if len(t) > 3 else 0, []).append(mm)
            res = []
            for g, ms in sorted(groups.items()):
                o = case.get("group_oper", ["and"] * 8)[g]
                res.append(np.logical_or.reduce(ms) if o == "or" else np.logical_and.reduce(ms))
            m = np.logical_or.reduce(res) if case.get("top_oper") == "or" else np.logical_and.reduce(res)
        if case.get("clusters"):
            m &= np.isin(hosts % 3, [int(c_[2:]) for c_ in case["clusters"]])
        if case.get("machine_ids"):
            m &= np.isin(hosts, [h for h in range(nh) if any(mids[h] == x for x in case["machine_ids"])])
        sel = np.nonzero(m)[0]
        if case.get("sort_col"):
            v = rows[sel, cols.index(case["sort_col"])]
            sel = sel[np.argsort(-v if case.get("sort_desc", True) else v, kind="stable")]
        sel = sel[:case.get("maxrecs", 1 << 30)]
        js = json.loads(eng.json_svcsumm_multihost(case.get("terms"), case.get("group_oper", ()), case.get("top_oper", "and"), case.get("sort_col"),
                                                   case.get("sort_desc", True), case.get("maxrecs", 1 << 30), case.get("machine_ids"), case.get("clusters"),
                                                   madid="cd" * 8, timestr="T"))
        assert list(js.keys()) == ["madid", "summstats"] and js["madid"] == "cd" * 8
        assert [r["host"] for r in js["summstats"]] == ["host%03d" % hosts[i] for i in sel], ci
        for r, i in zip(js["summstats"], sel):
            assert list(r.keys()) == ["parid", "host", "madid", "cluster"] + tj.SVCSUMM_COLS
            assert [r[c_] for c_ in cols] == rows[i].tolist() and r["cluster"] == "cl%d" % (hosts[i] % 3)
    eng.close()


def test_svcstate_scan_at_scale_top_1000_of_many_services():
    """10^6 services over 1 000 hosts: filter + exact top-1000 by a column equals numpy; the kernel time is printed (bench.py reports the
    10^7-service figure)"""
    import torch
    rng = np.random.default_rng(405)
    nh, sp = 1000, 1000
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    s_ = np.arange(sp)
    recs = []
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "c")
        eng.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))
    # one device-resident call for all hosts
    rec = np.concatenate([wire.synth_listener_states(rng, h, s_) for h in range(0, nh, 50)])  # 20 hosts' worth of distinct records ...
    rec = np.tile(rec, 50)                                                                  # ... repeated (values), with every service's own id
    rec["glob_id"] = np.concatenate([wire.glob_id(np.full(sp, h), s_) for h in range(nh)])
    rec["nqrys_5s"] = rng.integers(0, 1 << 22, len(rec))
    d = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()).cuda()
    off = torch.arange(0, len(rec) * 88, 88, dtype=torch.int32, device="cuda")
    hostl = torch.from_numpy(np.repeat(np.arange(nh, dtype=np.uint32), sp).view(np.int32).copy()).cuda()
    import ctypes as C
    eng.order()
    capi.check(eng.L.gys_ingest_listener_state_dev(eng.h, C.c_void_p(d.data_ptr()), C.c_void_p(off.data_ptr()), C.c_void_p(hostl.data_ptr()), len(rec)))
    eng.window_close()
    ok = rec["curr_state"] <= 5
    m = match(rec, [("qps5s", ">", 1000), ("p95resp5s", ">", 30)]) & ok
    v = col_values(rec[m], "qps5s")
    slot = np.arange(len(rec))
    order = np.lexsort((slot[m], -v))[:1000]
    eng.profile(True)
    eng.profile_reset()
    gs, gh, gr, nm = eng.svcstate_scan([("qps5s", ">", 1000), ("p95resp5s", ">", 30)], sort_col="qps5s", sort_desc=True, maxrecs=1000)
    prof = eng.profile_get()
    assert nm == int(m.sum()) and gs.tolist() == slot[m][order].tolist()
    print("svc_filter: %d services, %d matched, top-1000 in %.3f ms (kernels)" % (len(rec), nm, prof["svc_filter"][0]))
    # a name criterion over the whole registry (slot ranges on the host's cores from 2^16 services on): ids in registration order
    ids = eng.svc_ids_by_name("like", "^sv.$")
    assert len(ids) == nh * sp and (ids == rec["glob_id"]).all()
    assert len(eng.svc_ids_by_name("substr", "nginx")) == 0
    eng.close()


def test_submission_queue_tail_is_flushed_without_another_call():
    """calls that find GYS_RQ_INFLIGHT submissions on the GPU leave their events in the open batch; after a burst nobody calls again.
    The queue's flusher thread submits that tail once a submission has retired -- watched through gys_resp_queue_pending, the one entry
    point that does not flush the queue itself: after a burst from 16 threads the queue drains with no further call.  The state ends
    bit-identical to an engine fed the same calls one by one (that path is compared with the oracle in tests/test_gpu_round3.py)."""
    import threading
    import time
    from tests import helpers
    nh, sp, rounds, nthreads = 32, 10, 8, 16
    rng = np.random.default_rng(31)
    calls = {h: [helpers.make_resp_events(rng, h, 20000, sp) for _ in range(rounds)] for h in range(nh)}
    states = []
    for burst in (True, False):
        eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=1 << 22)
        info, gids = helpers.register_world(eng, None, range(nh), sp)
        if burst:
            start = threading.Barrier(nthreads)
            errs = []

            def worker(t):
                try:
                    start.wait()
                    for r in range(rounds):
                        for h in range(t, nh, nthreads):
                            eng.handle_resp_events(info[h][0], calls[h][r])
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex)

            th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not errs, errs
            left = eng.resp_queue_pending()
            t0 = time.time()
            while eng.resp_queue_pending() and time.time() - t0 < 5.0:  # no further call: only the flusher can submit what is left
                time.sleep(0.001)
            assert eng.resp_queue_pending() == 0, "the tail of the burst stayed in the queue"
            eng.sync()
            c = eng.counters()
            assert left == 0 or c["resp_tail_flushes"] >= 1, (left, c["resp_tail_flushes"])
            print("queue tail: %d events were left behind by the burst, %d flusher submission(s), %d calls in %d submissions" %
                  (left, c["resp_tail_flushes"], c["resp_calls_queued"], c["resp_submissions"]))
        else:
            for r in range(rounds):
                for h in range(nh):
                    eng.handle_resp_events(info[h][0], calls[h][r])
                    eng.sync()
        n = nh * sp
        gs, gc, gm = eng.export_tdigest(0, n)
        gn, gp = eng.export_tdigest_pending(0, n)
        h_all = eng.export_hist(1, 0, n)
        states.append((gs.tobytes(), gc.tobytes(), gm.tobytes(), gn.tobytes(), gp.tobytes(), h_all.tobytes(), eng.counters()["resp_events"]))
        eng.close()
    assert states[0] == states[1]


def test_conn_and_listener_state_calls_from_16_threads_are_combined_and_equal_sequential_calls(oracle):
    """gys_ingest_tcp_conn / gys_ingest_listener_state from 16 threads go through their submission queues (several parthas' messages per
    H2D copy + launch); the state must equal the same messages sent one by one from one thread -- registers, counters, per-host
    summaries, and the kept 88-byte state of every listener (a partha's messages keep their order: the last one wins)."""
    import threading
    from gyeeta_amd.engine import SketchEngine
    nh, sp, rounds, nthreads = 48, 40, 5, 16
    rng = np.random.default_rng(77)
    engs = [SketchEngine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False) for _ in range(2)]
    infos = [helpers.register_world(e, None, range(nh), sp)[0] for e in engs]
    conn_msgs, ls_msgs = {}, {}
    for h in range(nh):
        conn_msgs[h], ls_msgs[h] = [], []
        for r in range(rounds):
            n = int(rng.integers(900, 2049))
            rec = wire.synth_tcp_conns(rng, n, [h], sp, dup_frac=0.2, v6_frac=0.1)
            tails = [bytes(rng.integers(32, 127, int(k), dtype=np.uint8).tolist()) for k in rng.integers(0, 65, n) * (rng.random(n) < 0.2)]
            conn_msgs[h].append((wire.pack_variable(rec, tails), n))
            ls = wire.synth_listener_states(rng, h, np.arange(sp), delete_frac=0.02, bad_state_frac=0.02)
            ls_msgs[h].append((wire.pack_variable(ls, [b"x" * int(k) for k in rng.integers(0, 9, sp)]), sp))
    # sequential engine: one thread, message by message
    for h in range(nh):
        for r in range(rounds):
            engs[1].partha_tcp_conn_info(infos[1][h][0], *conn_msgs[h][r])
            engs[1].partha_listener_state(infos[1][h][0], *ls_msgs[h][r])
    errs = []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for r in range(rounds):
                for h in range(t, nh, nthreads):  # a partha's messages arrive in order on its connection
                    engs[0].partha_tcp_conn_info(infos[0][h][0], *conn_msgs[h][r])
                    engs[0].partha_listener_state(infos[0][h][0], *ls_msgs[h][r])
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    c0, c1 = engs[0].counters(), engs[1].counters()
    for k in ("conn_events", "conn_unknown_service", "conn_new", "conn_closed", "conn_closed_no_notify", "conn_client_side",
              "lstate_records", "lstate_missed", "lstate_errors", "lstate_deleted"):
        assert c0[k] == c1[k], k
    assert c0["conn_calls_queued"] == c0["lstate_calls_queued"] == nh * rounds == c1["conn_calls_queued"]
    assert 1 <= c0["conn_submissions"] <= nh * rounds and 1 <= c0["lstate_submissions"] <= nh * rounds
    assert 1 <= c1["conn_submissions"] <= nh * rounds  # (a lone caller's message goes out at once unless two submissions are still on the GPU)
    # kept states of every listener, before the window closes
    s0, h0, r0, n0 = engs[0].svcstate_scan(maxrecs=nh * sp + 10)  # (no sort column: by service slot)
    s1, h1, r1, n1 = engs[1].svcstate_scan(maxrecs=nh * sp + 10)
    assert n0 == n1 == len(s0) == len(s1) and n0 > 0.9 * nh * sp
    assert (s0 == s1).all() and (h0 == h1).all() and r0.tobytes() == r1.tobytes()
    for e in engs:
        e.window_close()
    assert (engs[0].export_hll() == engs[1].export_hll()).all()
    for w in (0, 1):
        assert (engs[0].export_cms(w) == engs[1].export_cms(w)).all()
    assert (engs[0].export_svc_counters() == engs[1].export_svc_counters()).all()
    for h in range(nh):
        assert engs[0].svcsumm(infos[0][h][0]).as_tuple() == engs[1].svcsumm(infos[1][h][0]).as_tuple()
    # a call of many messages' size bypasses the queue (behind what it holds) and is still counted and ingested
    big = wire.synth_tcp_conns(rng, 40000, list(range(nh)), sp, dup_frac=0.1)
    before = engs[0].counters()
    engs[0].partha_tcp_conn_info(infos[0][0][0], wire.pack_variable(big, None), len(big))
    after = engs[0].counters()
    assert after["conn_events"] - before["conn_events"] == len(big) and after["conn_calls_queued"] == before["conn_calls_queued"]
    for e in engs:
        e.close()


def test_service_name_criterion_resolves_to_service_ids():
    """gys_svc_ids_by_name: the reference's string comparators on the service name (CRITERION_ONE::match_str_criterian,
    common/gy_query_criteria.h:1335-1383: = / != whole name, substr / notsubstr memmem, like / notlike a regular expression matched
    anywhere, in / notin whole names) against python's own string operations; the ids then select rows of the filtered query."""
    import re
    eng = _engine(max_hosts=4, max_services=256, enable_tdigest=False)
    names = [b"nginx", b"postgres", b"post-worker", b"redis-server", b"java", b"x" * 16, b"envoy", b"post"]  # (a 16-byte name has no terminator)
    gid_name = {}
    rng = np.random.default_rng(5)
    for h in range(3):
        mid = wire.machine_id(h)
        eng.register_host(mid, "c0")
        for k, nm in enumerate(names):
            s = np.arange(4) + 4 * k
            gids = wire.glob_id(np.full(4, h), s)
            eng.register_listeners(mid, gids, wire.listener_netns(h, s), wire.listener_port(s), comm=nm)
            for g in gids:
                gid_name[int(g)] = nm.decode()
    cases = [("=", ["post"], lambda n: n == "post"), ("!=", ["post"], lambda n: n != "post"),
             ("substr", ["post"], lambda n: "post" in n), ("notsubstr", ["post"], lambda n: "post" not in n),
             ("like", ["^post.*r$"], lambda n: re.search("^post.*r$", n) is not None), ("notlike", ["^(nginx|envoy)$"], lambda n: re.search("^(nginx|envoy)$", n) is None),
             ("like", ["x{16}"], lambda n: re.search("x{16}", n) is not None),
             ("in", ["java", "redis-server", "nope"], lambda n: n in ("java", "redis-server")), ("notin", ["java", "post"], lambda n: n not in ("java", "post")),
             ("substr", ["a-name-longer-than-16-bytes"], lambda n: False)]
    for comp, pats, pred in cases:
        got = set(int(x) for x in eng.svc_ids_by_name(comp, pats))
        want = {g for g, n in gid_name.items() if pred(n)}
        assert got == want, (comp, pats, len(got), len(want))
    with pytest.raises(capi.GysError):
        eng.svc_ids_by_name("like", ["(unclosed"])
    with pytest.raises(capi.GysError):
        eng.svc_ids_by_name("<", ["post"])
    # the ids select rows of the filtered multi-host query: states of every listener, then { name substr 'post' and qps5s >= 0 }
    for h in range(3):
        ls = wire.synth_listener_states(rng, h, np.arange(4 * len(names)))
        eng.partha_listener_state(wire.machine_id(h), wire.pack_variable(ls, None), len(ls))
    ids = eng.svc_ids_by_name("substr", "post")
    slots, hosts, recs, nm = eng.svcstate_scan(terms=[("qps5s", ">=", 0)], maxrecs=1000, svcids=ids)
    assert nm == len(slots) > 0 and set(int(g) for g in recs["glob_id"]) <= set(int(x) for x in ids)
    _, _, recs_all, nall = eng.svcstate_scan(terms=[("qps5s", ">=", 0)], maxrecs=1000)
    assert set(int(g) for g in recs["glob_id"]) == {int(g) for g in recs_all["glob_id"] if "post" in gid_name[int(g)]}
    # the same comparators on the host name -> machine ids for the filter
    hn = {0: "db-east-1", 1: "web-east-2", 2: "db-west-1"}
    for h, name in hn.items():
        eng.set_host_name(wire.machine_id(h), name)
    for comp, pats, pred in (("substr", ["db-"], lambda n: "db-" in n), ("like", ["east-[0-9]$"], lambda n: re.search("east-[0-9]$", n) is not None),
                             ("!=", ["web-east-2"], lambda n: n != "web-east-2"), ("in", ["db-west-1", "x"], lambda n: n in ("db-west-1", "x")),
                             ("notsubstr", ["e"], lambda n: "e" not in n)):
        got = set(eng.machine_ids_by_hostname(comp, pats))
        assert got == {bytes(wire.machine_id(h)) for h, n in hn.items() if pred(n)}, (comp, pats)
    # `like` is an automaton search (the reference: RE2): patterns a backtracking matcher needs exponential time for answer at once, on a
    # host name of the longest kind; RE2's syntax ((?i), [[:digit:]], \\z) is taken, what RE2 rejects (back-references, look-around) is rejected
    import time
    eng.set_host_name(wire.machine_id(1), "a" * 250 + "!" + "tail-that-is-cut-at-255-bytes")
    t0 = time.perf_counter()
    for pat in ("(a+)+$", "(a|aa)+$", "(a*)*c", "(.*a){25}x"):
        assert eng.machine_ids_by_hostname("like", [pat]) == []
    assert time.perf_counter() - t0 < 0.5
    assert set(eng.machine_ids_by_hostname("like", ["(?i)^A{250}!tail$"])) == {bytes(wire.machine_id(1))}  # (cut at 255 bytes: 250 + '!' + 'tail')
    assert set(eng.machine_ids_by_hostname("like", ["^db-[[:alpha:]]+-[[:digit:]]\\z"])) == {bytes(wire.machine_id(0)), bytes(wire.machine_id(2))}
    for bad in ("(a)\\1", "(?=db)", "\\pL+", "a{2000}"):
        with pytest.raises(capi.GysError):
            eng.machine_ids_by_hostname("like", [bad])
    eng.set_host_name(wire.machine_id(1), hn[1])
    mids = eng.machine_ids_by_hostname("substr", "db-")
    _, hosts_sel, _, nsel = eng.svcstate_scan(maxrecs=1000, machine_ids=mids)
    assert nsel > 0 and set(int(x) for x in hosts_sel) == {0, 2}
    eng.close()
