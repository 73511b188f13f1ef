"""Query results as JSON in the reference's web shapes: key names AND order follow common/gy_json_field_maps.h
(json_db_svcsumm_arr :1396-1416, json_db_svcstate_arr :1102-1135, json_db_clusterstate_arr :2162-2180), envelopes follow
web_curr_listener_summ / web_curr_listener_state / web_curr_clusterstate; values are checked against the ingested records."""
import json

import numpy as np
import pytest

from gyeeta_amd import wire

pytestmark = pytest.mark.gpu

SVCSUMM_COLS = ["time", "nidle", "ngood", "nok", "nbad", "nsevere", "ndown", "totqps", "totaconn", "totkbin", "totkbout", "totsererr", "nsvc", "nactive"]
SVCSTATE_COLS = ["time", "svcid", "name", "qps5s", "nqry5s", "resp5s", "p95resp5s", "p95resp5m", "nconns", "nactive", "nprocs", "kbin15s", "kbout15s",
                 "sererr", "clierr", "delayus", "cpudelus", "iodelus", "vmdelus", "usercpu", "syscpu", "rssmb", "nissue", "state", "issue", "ishttp", "desc"]
CLUSTER_COLS = ["time", "cluster", "nhosts", "nprocissue", "nprochosts", "nproc", "nlistissue", "nlisthosts", "nlisten", "totqps", "svcnetmb",
                "ncpuissue", "nmemissue"]
STATES = ["Idle", "Good", "OK", "Bad", "Severe", "Down"]


def test_json_shapes_and_values():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    from gyeeta_amd.engine import SketchEngine
    rng = np.random.default_rng(3)
    eng = SketchEngine(max_hosts=4, max_services=64, enable_tdigest=False)
    hosts = {0: ("web-1", "prod"), 1: ('db "one"\\x', "prod"), 2: ("idle-host", "staging")}
    recs = {}
    for h, (name, cluster) in hosts.items():
        mid = wire.machine_id(h)
        eng.register_host(mid, cluster)
        eng.set_host_name(mid, name)
        s = np.arange(5)
        eng.register_listeners(mid, wire.glob_id(np.full(5, h), s), wire.listener_netns(h, s), wire.listener_port(s), comm=b"svc-%d" % h)
    for h in (0, 1):
        ls = wire.synth_listener_states(rng, h, np.arange(5))
        ls["tasks_delay_usec"] = ls["tasks_cpudelay_usec"] + ls["tasks_blkiodelay_usec"] + rng.integers(0, 50, 5)
        ls["is_http_svc"] = rng.integers(0, 2, 5)
        eng.partha_listener_state(wire.machine_id(h), ls.tobytes(), 5)
        eng.handle_host_state(wire.machine_id(h), ntasks_issue=h, ntasks=40 + h, nlisten_issue=1, nlisten=5, cpu_issue=h, mem_issue=0)
        recs[h] = ls
    eng.window_close()
    madid, ts = "00aa11bb22cc33dd", "2026-01-01T00:00:05+0000"
    for h in (0, 1):
        mid = wire.machine_id(h)
        d = json.loads(eng.json_svcsumm(mid, madid, ts))
        assert list(d.keys()) == ["madid", "summstats", "hostinfo"] and d["madid"] == madid
        row = d["summstats"][0]
        assert list(row.keys()) == SVCSUMM_COLS and row["time"] == ts
        summ = eng.svcsumm(mid).as_tuple()
        assert [row[k] for k in SVCSUMM_COLS[1:]] == list(summ)
        a, b = np.frombuffer(bytes(mid[:8]), "<u8")[0], np.frombuffer(bytes(mid[8:]), "<u8")[0]
        assert d["hostinfo"] == {"parid": "%016x%016x" % (a, b), "host": hosts[h][0], "madid": madid, "cluster": hosts[h][1]}
        d = json.loads(eng.json_svcstate(mid, madid, ts))
        assert list(d.keys()) == ["madid", "svcstate", "hostinfo"] and len(d["svcstate"]) == 5
        ls = recs[h]
        for i, row in enumerate(d["svcstate"]):
            assert list(row.keys()) == SVCSTATE_COLS
            r = ls[i]
            nq = int(r["nqrys_5s"])
            exp = {"time": ts, "svcid": "%016x" % int(r["glob_id"]), "name": "svc-%d" % h, "qps5s": nq // 5, "nqry5s": nq,
                   "resp5s": int(r["total_resp_5sec"]) // (nq if nq else 1), "p95resp5s": int(r["p95_5s_resp_ms"]), "p95resp5m": int(r["p95_5min_resp_ms"]),
                   "nconns": int(r["nconns"]), "nactive": int(r["nconns_active"]), "nprocs": int(r["ntasks"]), "kbin15s": int(r["curr_kbytes_inbound"]),
                   "kbout15s": int(r["curr_kbytes_outbound"]), "sererr": int(r["ser_errors"]), "clierr": int(r["cli_errors"]),
                   "delayus": int(r["tasks_delay_usec"]), "cpudelus": int(r["tasks_cpudelay_usec"]), "iodelus": int(r["tasks_blkiodelay_usec"]),
                   "vmdelus": max(0, int(r["tasks_delay_usec"]) - int(r["tasks_cpudelay_usec"]) - int(r["tasks_blkiodelay_usec"])),
                   "usercpu": int(r["tasks_user_cpu"]), "syscpu": int(r["tasks_sys_cpu"]), "rssmb": int(r["tasks_rss_mb"]), "nissue": int(r["ntasks_issue"]),
                   "state": STATES[int(r["curr_state"])], "issue": int(r["curr_issue"]), "ishttp": bool(r["is_http_svc"]), "desc": ""}
            assert row == exp
    # a host that reported nothing in the window: empty svcstate array, zero summary
    d = json.loads(eng.json_svcstate(wire.machine_id(2), madid, ts))
    assert d["svcstate"] == [] and d["hostinfo"]["host"] == "idle-host"
    d = json.loads(eng.json_clusterstate("ffee" * 4, ts))
    assert list(d.keys()) == ["shyamaid", "clusterstate"] and d["shyamaid"] == "ffee" * 4
    assert [c["cluster"] for c in d["clusterstate"]] == ["prod"]  # staging had no host state in the window
    row = d["clusterstate"][0]
    assert list(row.keys()) == CLUSTER_COLS
    cs = eng.clusterstate("prod").as_tuple()
    assert [row[k] for k in CLUSTER_COLS[2:]] == list(cs) and row["nhosts"] == 2 and row["nproc"] == 81 and row["ncpuissue"] == 1
    # buffer too small -> GYS_ERR_NOMEM with the needed size reported
    import ctypes as C
    from gyeeta_amd import capi
    need = C.c_size_t()
    small = C.create_string_buffer(8)
    rc = eng.L.gys_json_clusterstate(eng.h, b"x", b"", small, 8, C.byref(need))
    assert rc == capi.ERR_NOMEM and need.value > 8
    eng.close()


def test_json_toplisteners_single_and_multi_host():
    """web_curr_top_listeners (server/gy_mnodehandle.cc:2706-3190): per host the 10 best services of the last window by each of the four
    LISTEN_TOPN orders; the multi-host form merges every host's queues into 50 slots per kind.  Expected sets are computed from the
    ingested records with numpy (metric descending, ties by lower slot = the engine's deterministic rule)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    from gyeeta_amd import capi
    from gyeeta_amd.engine import SketchEngine
    rng = np.random.default_rng(11)
    nh, sp = 9, 30
    eng = SketchEngine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    recs = {}
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "cl%d" % (h % 2))
        eng.set_host_name(mid, "host-%d" % h)
        s = np.arange(sp)
        eng.register_listeners(mid, wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s), comm=b"svc")
    for h in range(nh - 1):  # the last host reports nothing
        ls = wire.synth_listener_states(rng, h, np.arange(sp))
        ls["curr_state"] = rng.integers(0, 6, sp)
        ls["tasks_delay_usec"] = rng.integers(0, 5000, sp)
        ls["tasks_cpudelay_usec"] = 0
        ls["tasks_blkiodelay_usec"] = 0
        ls["nconns_active"] = rng.integers(0, 40, sp)
        ls["curr_kbytes_inbound"] = rng.integers(0, 300, sp)
        ls["curr_kbytes_outbound"] = rng.integers(0, 300, sp)
        eng.partha_listener_state(wire.machine_id(h), ls.tobytes(), sp)
        recs[h] = ls
    eng.window_close()

    def expected(h, kind):
        r = recs[h]
        slot = h * sp + np.arange(sp)
        if kind == 0:
            ok = r["curr_state"] > 2
            metric = (r["curr_state"].astype(np.uint64) << np.uint64(32)) | r["tasks_delay_usec"].astype(np.uint64)
        elif kind == 1:
            ok, metric = r["nqrys_5s"] >= 5, r["nqrys_5s"].astype(np.uint64)
        elif kind == 2:
            ok, metric = r["nconns_active"] >= 1, r["nconns_active"].astype(np.uint64)
        else:
            metric = r["curr_kbytes_inbound"].astype(np.uint64) + r["curr_kbytes_outbound"].astype(np.uint64)
            ok = metric > 0
        order = sorted((i for i in range(sp) if ok[i]), key=lambda i: (-int(metric[i]), int(slot[i])))[:10]
        return [(int(metric[i]), int(slot[i]), "%016x" % int(r["glob_id"][i])) for i in order]

    names = ["topissue", "topqps", "topactconn", "topnet"]
    madid, ts = "0123456789abcdef", "2026-01-01T00:00:05+0000"
    d = json.loads(eng.json_toplisteners(wire.machine_id(3), 15 | 16, madid, ts))
    assert list(d.keys()) == ["madid"] + names + ["summstats", "hostinfo"]
    for kind, nm in enumerate(names):
        want = expected(3, kind)
        assert [e["svcid"] for e in d[nm]] == [w[2] for w in want], nm
        for e in d[nm]:
            assert list(e.keys()) == SVCSTATE_COLS + ["ip", "port"] and e["time"] == ts
    assert d["summstats"]["nsvc"] == sp and d["hostinfo"]["host"] == "host-3"
    e0 = d["topqps"][0]
    i0 = int(np.argmax(recs[3]["nqrys_5s"]))
    assert e0["nqry5s"] == int(recs[3]["nqrys_5s"][i0]) and e0["port"] == int(wire.listener_port(np.arange(sp))[i0])
    # only the requested arrays are sent; no criteria at all is an error (reference: ERR_INVALID_REQUEST)
    d = json.loads(eng.json_toplisteners(wire.machine_id(3), 2, madid, ts))
    assert list(d.keys()) == ["madid", "topqps", "hostinfo"]
    with pytest.raises(capi.GysError):
        eng.json_toplisteners(wire.machine_id(3), 16, madid, ts)
    # multi-host: union of the hosts' top-10 queues, best 50 per kind, entries carry the host identity
    d = json.loads(eng.json_toplisteners(None, 15 | 16, madid, ts))
    assert list(d.keys()) == ["madid"] + names + ["summstats"]
    for kind, nm in enumerate(names):
        union = sorted((w for h in range(nh - 1) for w in expected(h, kind)), key=lambda w: (-w[0], w[1]))[:50]
        assert [e["svcid"] for e in d[nm]] == [w[2] for w in union], nm
        assert len(d[nm]) <= 50
        for e, w in zip(d[nm], union):
            h = w[1] // sp
            assert list(e.keys())[:4] == ["parid", "host", "madid", "cluster"] and e["host"] == "host-%d" % h and e["cluster"] == "cl%d" % (h % 2)
    assert d["summstats"]["nsvc"] == (nh - 1) * sp
    eng.close()
