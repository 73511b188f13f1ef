"""Round-3 parity additions:
  * the host-pointer response path under concurrency: 16 threads (the reference's MAX_L2_MISC_THREADS, server/gy_mconnhdlr.h:60) call
    gys_ingest_resp_events at the same time; the library combines pending calls into common submissions (submission queue) and the
    resulting state -- histograms, digests, buffered values, HLL, Count-Min -- equals the oracle fed call by call;
  * the staging ring of the other host-pointer calls hands its slots out oldest first (no wait for the GPU unless the ring wrapped)."""
import threading

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


def test_resp_calls_from_16_threads_are_combined_and_bit_exact(torch_mod, oracle):
    nh, sp, rounds, per_call = 48, 20, 6, 3000
    eng = _engine(max_hosts=nh, max_services=nh * sp, max_batch_events=1 << 20)
    orc = oracle.OracleEngine(nh * sp)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    rng = np.random.default_rng(2024)
    nthreads = 16
    # every thread owns hosts t, t + 16, t + 32 and sends `rounds` calls per host, in order (a partha's messages arrive in order on its
    # connection; different parthas interleave freely)
    calls = {h: [helpers.make_resp_events(rng, h, per_call + 37 * (h % 5), sp) for _ in range(rounds)] for h in range(nh)}
    errs = []
    start = threading.Barrier(nthreads)

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
    eng.sync()
    for h in range(nh):  # the oracle sees every call on its own, per host in call order (hosts are independent)
        for r in range(rounds):
            orc.resp_batch(calls[h][r].tobytes(), [info[h][1]], [0])
    n = orc.nsvc
    helpers.assert_hist_equal(eng.export_hist(0, 0, n), orc.hist(), n)
    gs, gc, gm = eng.export_tdigest(0, n)
    os_, oc, om = orc.td_arrays()
    assert (gs == os_).all() and (gc == oc).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, n)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    c = eng.counters()
    assert c["resp_calls_queued"] == nh * rounds
    assert 1 <= c["resp_submissions"] <= c["resp_calls_queued"]
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all() and (eng.export_cms(0) == orc.cms()).all()
    orc.window_clear(clear_hist=True)
    # a lone caller with an idle GPU is submitted at once (no added latency); whatever the queue still holds goes out with the next
    # entry point that reads or closes state
    eng.sync()
    before = eng.counters()
    eng.handle_resp_events(info[0][0], calls[0][0])
    orc.resp_batch(calls[0][0].tobytes(), [info[0][1]], [0])
    assert eng.counters()["resp_submissions"] - before["resp_submissions"] == 1
    for r in range(1, 4):
        eng.handle_resp_events(info[0][0], calls[0][r])  # the same host again: every call its own segment of its own batch, in call order
        orc.resp_batch(calls[0][r].tobytes(), [info[0][1]], [0])
    helpers.assert_hist_equal(eng.export_hist(0, 0, n), orc.hist(), n)
    gs, gc, gm = eng.export_tdigest(0, n)
    os_, oc, om = orc.td_arrays()
    assert (gs == os_).all() and (gc == oc).all() and (gm == om).all()
    eng.close()


def test_staging_ring_is_fifo(torch_mod, oracle):
    """gys_ingest_listener_state / _tcp_conn copy into a ring of 16 pinned slots; a slot is taken oldest first, so a burst shorter
    than the ring never waits for the GPU (gys_counters.stage_waits)"""
    nh, sp = 4, 16
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    info, gids = helpers.register_world(eng, None, range(nh), sp)
    rng = np.random.default_rng(5)
    recs = [wire.synth_listener_states(rng, h, np.arange(sp)) for h in range(nh)]
    eng.sync()
    w0 = eng.counters()["stage_waits"]
    for i in range(12):
        h = i % nh
        eng.partha_listener_state(info[h][0], recs[h].tobytes(), sp)
    eng.sync()
    assert eng.counters()["stage_waits"] == w0
    eng.close()


# ------------------------------------------------------------------------------------------------ in-library exchange at nranks = 2
def _fake_rank(rank, q_uid, q_res):
    """one rank of test_window_close_rccl_two_ranks: its shard of the hosts, the in-library exchange, what it observes afterwards"""
    import ctypes as C
    import os
    import torch
    from gyeeta_amd import capi
    from gyeeta_amd.engine import SketchEngine, mid_buf
    from tests.test_gpu_round2 import NH, SP, _feed, _observe
    try:
        torch.cuda.set_device(0)
        L = capi.load()
        glob = C.CDLL(os.environ["GYS_RCCL_LIB"])  # (the same handle the library's dlopen got)
        glob.fakerccl_allreduce_calls.restype = C.c_uint64
        glob.fakerccl_allgather_calls.restype = C.c_uint64
        mine = [h for h in range(NH) if L.gys_shard_of(mid_buf(wire.machine_id(h)), 2) == rank]
        eng = SketchEngine(max_hosts=NH, max_services=NH * SP, max_clusters=4, max_batch_events=1 << 14, rank=rank, nranks=2, device=0)
        _feed(eng, mine)
        if rank == 0:
            uid = bytes(eng.rccl_unique_id())
            q_uid.put(uid)
        else:
            uid = q_uid.get(timeout=120)
        eng.join_rccl(uid)
        eng.window_close_rccl(tusec=5_000_000)
        obs = _observe(eng)
        out = torch.zeros(C.sizeof(capi.TDigestSlab), dtype=torch.uint8, device="cuda")
        capi.check(L.gys_tdigest_global_rccl(eng.h, eng.comm, C.c_void_p(out.data_ptr())))
        eng.sync()
        merged = out.cpu().numpy().tobytes()
        dev_l, _ = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
        local = dev_l.cpu().numpy().tobytes()
        calls = (int(glob.fakerccl_allreduce_calls()), int(glob.fakerccl_allgather_calls()))
        eng.leave_rccl()
        eng.close()
        q_res.put((rank, "ok", obs, merged, local, calls, len(mine)))
    except BaseException as ex:  # noqa: BLE001 -- reported to the parent
        import traceback
        q_res.put((rank, "error: " + "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-2000:]))


def test_window_close_rccl_two_ranks(torch_mod, oracle):
    """gys_window_close_rccl and gys_tdigest_global_rccl with TWO ranks: two processes on the one GPU of the box, each
    owning its shard of the hosts; the RCCL entry points the library calls are served by tests/cpp/fakerccl ($GYS_RCCL_LIB; RCCL itself
    refuses two ranks on one device), which executes every all-reduce / all-gather with exactly the count, datatype, operator and
    pointers the library passed.  Both ranks end with the registers, Count-Min tables, all-service histogram and cluster rows of a
    single-rank engine fed all hosts, and with the same global digest = the oracle's roll-up of the two ranks' slabs."""
    import ctypes as C
    import os
    import subprocess
    import torch.multiprocessing as mp
    from gyeeta_amd import capi
    from tests.test_gpu_round2 import NH, SP, _feed, _observe
    here = os.path.dirname(os.path.abspath(__file__))
    fake = os.path.join(here, "cpp", "fakerccl", "libfakerccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               os.path.join(here, "cpp", "fakerccl", "fakerccl.cc"), "-o", fake, "-L/opt/rocm/lib", "-lamdhip64", "-lrt",
                               "-Wl,-rpath,/opt/rocm/lib"])
    ctx = mp.get_context("spawn")
    q_uid, q_res = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_fake_rank, args=(r, q_uid, q_res)) for r in range(2)]
    old = os.environ.get("GYS_RCCL_LIB")
    os.environ["GYS_RCCL_LIB"] = fake  # the library binds RCCL with dlopen($GYS_RCCL_LIB) (gys_engine.hip: rccl_api); inherited by the children
    try:
        for p in procs:
            p.start()
    finally:
        if old is None:
            del os.environ["GYS_RCCL_LIB"]
        else:
            os.environ["GYS_RCCL_LIB"] = old
    res = sorted(q_res.get(timeout=400) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] == "ok", r[1]
    for p in procs:
        assert p.exitcode == 0
    single = _engine(max_hosts=NH, max_services=NH * SP, max_clusters=4, max_batch_events=1 << 14)
    _feed(single, range(NH))
    single.window_close(tusec=5_000_000)
    want = _observe(single)
    assert res[0][6] > 0 and res[1][6] > 0 and res[0][6] + res[1][6] == NH
    names = ["hll", "cms32", "cms64", "global histogram", "cluster state", "distinct flows"]
    for r in range(2):
        assert res[r][5][0] >= 4 and res[r][5][1] >= 1, "the library's collectives did not go through the stand-in"
        for name, got, exp in zip(names, res[r][2], want):
            assert got == exp, f"rank {r}: {name} differs from the single-rank run"
    # global digest: identical on both ranks, equal to the oracle's roll-up of the two local slabs
    assert res[0][3] == res[1][3]
    tot, locs = 0, []
    for r in range(2):
        loc = np.frombuffer(res[r][4], dtype=single.SLAB_DT)[0]
        o1 = oracle.TD64()
        o1.sum[:] = loc["sum"].tolist()
        o1.cnt[:] = loc["cnt"].tolist()
        o1.vmin, o1.vmax = int(loc["vmin"]), int(loc["vmax"])
        locs.append(o1)
        tot += int(loc["cnt"].sum())
    d = oracle.rollup_slabs(locs)
    got = np.frombuffer(res[0][3], dtype=single.SLAB_DT)[0]
    assert int(got["cnt"].sum()) == tot > 0
    assert (got["sum"] == np.array(d.sum[:], dtype=np.int64)).all() and (got["cnt"] == np.array(d.cnt[:], dtype=np.uint64)).all()
    assert int(got["vmin"]) == d.vmin and int(got["vmax"]) == d.vmax
    single.close()


# ------------------------------------------------------------------------------------------------ the per-listener 5-s scan (row a9)
def test_listener_state_scan_from_engine_state_end_to_end(torch_mod, oracle):
    """gys_scan_listener_state_dev (TCP_SOCK_HANDLER::listener_stats_update + the data-parallel part of TCP_LISTENER::get_curr_state,
    common/gy_socket_stat.cc:4044-4365, :2030-2143): raw response events in, one LISTENER_STATE_NOTIFY + one scan record per service out
    of the engine's own state, equal field by field to the oracle's restatement (oracle/gy_oracle_lscan.c) driven by the folly-style
    ring oracle, the oracle's CONN_BITMAP and its QPS / active-connection histograms; the records are then fed to
    gys_ingest_listener_state_dev, so that raw events -> gys_json_svcstate / the host summary is one path end to end."""
    import ctypes as C
    import json
    from gyeeta_amd import capi
    from tests.test_gpu_levels import T0, RingOracle
    rng = np.random.default_rng(909)
    nh, sp = 3, 7
    nsvc = nh * sp
    eng = _engine(max_hosts=4, max_services=32, max_batch_events=1 << 15, enable_tdigest=True, enable_levels=True)
    orc_win = oracle.OracleEngine(32, enable_td=False)  # cleared at every close: the closing window's histograms and CONN_BITMAP rows
    info, gids = helpers.register_world(eng, orc_win, range(nh), sp)
    ring = RingOracle(oracle, nsvc)
    L = oracle.lib()
    qps_h = [oracle.Hist() for _ in range(nsvc)]
    act_h = [oracle.Hist() for _ in range(nsvc)]
    for s in range(nsvc):
        L.gyo_hist_init(C.byref(qps_h[s]), oracle.KINDS["SEMI_LOG_HASH_LO"])
        L.gyo_hist_init(C.byref(act_h[s]), oracle.KINDS["HASH_1_3000"])
    mult = 1.0
    steps = [5] * 8 + [30, 5, 5, 301, 5, 5, 6, 5]
    t = T0
    for w, dt in enumerate(steps):
        t += dt
        for h in range(nh):
            if w % 5 == 3 and h == 1:
                continue  # a silent host this window: its services report zero queries
            n = int(rng.integers(50, 3000))
            ev = helpers.make_resp_events(rng, h, n, int(rng.integers(2, sp + 1)), lat_mu=1.5 + 0.25 * (w % 9))
            eng.handle_resp_events(info[h][0], ev)
            orc_win.resp_batch(ev.tobytes(), [info[h][1]], [0])
        win = np.array(orc_win.hist()[:nsvc])
        rows = orc_win.bitmap()[:nsvc]
        eng.window_close(t * 1_000_000)
        ring.close(t, win)
        dev_notify, notify, scan = eng.scan_listener_state(t * 1_000_000, qps_multiple=mult, diffsec=dt)
        for s in range(nsvc):
            hc = oracle.MLHist.from_buffer_copy(ring.h[s])
            L.gyo_mlh_flush(C.byref(hc), t)
            want_n = np.zeros(88, dtype=np.uint8)
            want = oracle.ListenerScan()
            rr = np.ascontiguousarray(rows[s])
            L.gyo_listener_scan_one(C.byref(hc), C.byref(qps_h[s]), C.byref(act_h[s]), oracle.ptr(rr, oracle.u16p), int(gids[s // sp][s % sp]),
                                    mult, dt, oracle.ptr(want_n, oracle.u8p), C.byref(want))
            got = scan[s]
            assert int(got["glob_id"]) == want.glob_id
            for f in ("tcount", "tsum", "p95_ms", "p99_ms", "p25_ms"):
                assert got[f].tolist() == list(getattr(want, f)), (w, s, f, got[f].tolist(), list(getattr(want, f)))
            for f in ("last_qps", "curr_qps", "qps_p95", "qps_p25", "act_p95", "act_p25", "b5", "b300", "b5day", "nconn_active"):
                assert int(got[f]) == getattr(want, f), (w, s, f, int(got[f]), getattr(want, f))
            assert got["nactive_conn_arr"].tolist() == list(want.nactive_conn_arr), (w, s)
            assert notify[s].tobytes() == want_n.tobytes(), (w, s)
        assert (scan["tcount"][:, 0] == win[:, 15, 0]).all()  # nqrys_5s_ of a service = the response events of its closed window
        # feed the engine's own records back: host roll-up + top-N + the QPS / active-connection samples (k_lstate_ingest)
        off = eng.torch.arange(0, nsvc * 88, 88, dtype=eng.torch.int32, device="cuda")
        hostl = eng.torch.from_numpy(np.repeat(np.array([info[h][1] for h in range(nh)], dtype=np.int32), sp)).cuda()
        capi.check(eng.L.gys_ingest_listener_state_dev(eng.h, C.c_void_p(dev_notify.data_ptr()), C.c_void_p(off.data_ptr()), C.c_void_p(hostl.data_ptr()), nsvc))
        for s in range(nsvc):  # TCP_LISTENER histograms take one sample per record: nqrys_5s_ / 5 and nconns_active_ (k_lstate_ingest)
            L.gyo_hist_add(C.byref(qps_h[s]), int(notify["nqrys_5s"][s]) // 5)
            L.gyo_hist_add(C.byref(act_h[s]), int(notify["nconns_active"][s]))
        orc_win.window_clear(clear_hist=True)
    # the records of the last scan are what the web query shows after the next close
    eng.window_close((t + 5) * 1_000_000)
    for h in range(nh):
        js = json.loads(eng.json_svcstate(info[h][0]))
        by_id = {r["svcid"]: r for r in js["svcstate"]}
        for k in range(sp):
            s = h * sp + k
            r = by_id["%016x" % int(gids[h][k])]
            assert r["qps5s"] == int(notify["nqrys_5s"][s]) // 5 and r["p95resp5s"] == int(notify["p95_5s_resp_ms"][s])
            assert r["nactive"] == int(notify["nconns_active"][s])
    eng.close()


# ------------------------------------------------------------------------------------------------ hosts with more than 2048 listeners
@pytest.mark.parametrize("resp_path", [2, 3], ids=["hostlocal", "hostsplit"])
def test_many_listener_hosts_stay_on_the_host_local_path(torch_mod, oracle, resp_path):
    """a host whose listener table outgrows one LDS sub-table (2048 listeners) is cut into parts -- one workgroup per part, each
    resolving only its part's events -- instead of falling back to the general front end (VERDICT r2 missing 5; the reference's
    listener table has no such limit, common/gy_socket_stat.cc:1554-1677).  Registration crosses the 2048 and the 4096 mark between
    batches (1 -> 2 -> 4 parts with state in place); a small host shares the batches; results equal the oracle's bit for bit."""
    from gyeeta_amd import capi
    torch = torch_mod
    rng = np.random.default_rng(2048 + resp_path)
    big, small = 0, 1
    nbig, nsmall = 5200, 30
    eng = _engine(max_hosts=4, max_services=8192, max_batch_events=1 << 19, resp_path=resp_path)
    orc = oracle.OracleEngine(8192)
    mids = {h: wire.machine_id(h) for h in (big, small)}
    slots = {h: eng.register_host(mids[h], "c") for h in (big, small)}
    s_small = np.arange(nsmall)
    eng.register_listeners_np(mids[small], wire.glob_id(np.full(nsmall, small), s_small), wire.listener_netns(small, s_small), wire.listener_port(s_small))
    for i in range(nsmall):
        orc.register(slots[small], int(wire.glob_id(small, i)), int(wire.listener_netns(small, s_small)[i]), int(wire.listener_port(s_small)[i]))
    have = 0
    for upto in (1500, 2600, 4100, nbig):  # one table; two parts; four parts; four parts, fuller
        s = np.arange(have, upto)
        g, ns, pt = wire.glob_id(np.full(len(s), big), s), wire.listener_netns(big, s), wire.listener_port(s)
        eng.register_listeners_np(mids[big], g, ns, pt)
        for i in range(len(s)):
            orc.register(slots[big], int(g[i]), int(ns[i]), int(pt[i]))
        have = upto
        for rnd in range(2):
            # events over the listeners registered so far plus some not registered yet; the big host gets enough for multi-tile segments
            evb = helpers.make_resp_events(rng, big, 150000, min(nbig, have + 40))
            evs = helpers.make_resp_events(rng, small, 4000, nsmall)
            buf = helpers.concat_events([evb, evs])
            segs = (capi.RespSeg * 2)()
            segs[0].host_slot, segs[0].first_event = slots[big], 0
            segs[1].host_slot, segs[1].first_event = slots[small], len(evb)
            d = torch.from_numpy(buf.view(np.uint8).copy()).cuda()
            eng.handle_resp_events_dev(segs, d.data_ptr(), len(buf))
            eng.sync()
            orc.resp_batch(buf.tobytes(), [slots[big], slots[small]], [0, len(evb)])
    n = orc.nsvc
    helpers.assert_hist_equal(eng.export_hist(0, 0, n), orc.hist(), n)
    assert (eng.export_conn_bitmap(0, n) == orc.bitmap()).all()
    gs, gc, gm = eng.export_tdigest(0, n)
    os_, oc, om = orc.td_arrays()
    assert (gc == oc).all() and (gs == os_).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, n)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    c, oc_ = eng.counters(), orc.counters()
    assert c["resp_batches_general"] == 0, "a many-listener host must not push the batch onto the general front end"
    assert c["resp_events"] == oc_["events"] and c["resp_dropped_range"] == oc_["dropped_range"] and c["resp_dropped_nolistener"] == oc_["dropped_nolistener"] > 0
    if resp_path == 3:
        assert c["resp_batches_host_split"] > 0
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all() and (eng.export_cms(0) == orc.cms()).all()
    gh = eng.export_global_hist()
    oh, omax = orc.ghist()
    assert [gh.stats[i].count for i in range(15)] == oh[:15, 0].tolist() and gh.total_count == oh[15, 0] and gh.max_val_seen == omax
    eng.close()


# ------------------------------------------------------------------------------------------------ levels without the 5-s level
def test_levels_mode2_without_the_5s_level(torch_mod, oracle):
    """gys_config.enable_levels = 2: the 300-s / 5-day / all-time levels equal the folly-style ring oracle at every close and between
    closes although a close only touches the services when it crosses a ring boundary; level 0 and the listener scan are refused"""
    from gyeeta_amd import capi
    from tests.test_gpu_levels import T0, RingOracle, _check_levels
    rng = np.random.default_rng(5)
    nh, sp = 2, 6
    nsvc = nh * sp
    eng = _engine(max_hosts=4, max_services=32, max_batch_events=1 << 14, enable_tdigest=True, enable_levels=2)
    orc_win = oracle.OracleEngine(32, enable_td=False)
    orc_all = oracle.OracleEngine(32, enable_td=False)
    info, gids = helpers.register_world(eng, orc_win, range(nh), sp)
    helpers.register_world(None, orc_all, range(nh), sp)
    ring = RingOracle(oracle, nsvc)
    steps = [5] * 14 + [7, 3, 5, 5, 40, 5, 5, 301, 5, 5, 5, 43200, 5, 5] + [5] * 8
    t = T0
    for w, dt in enumerate(steps):
        t += dt
        for h in range(nh):
            if rng.random() < 0.8:
                ev = helpers.make_resp_events(rng, h, int(rng.integers(1, 600)), int(rng.integers(1, sp + 1)), lat_mu=2.0 + 0.1 * (w % 20))
                eng.handle_resp_events(info[h][0], ev)
                for o in (orc_win, orc_all):
                    o.resp_batch(ev.tobytes(), [info[h][1]], [0])
        win = np.array(orc_win.hist()[:nsvc])
        eng.window_close(t * 1_000_000)
        ring.close(t, win)
        allmax = np.array(orc_all.hist()[:nsvc])[:, 15, 1]
        _check_levels(eng, ring, t, nsvc, [1, 2, 3], allmax)
        if w % 3 == 1:
            _check_levels(eng, ring, t + int(rng.integers(1, 300)), nsvc, [1, 2, 3], allmax)
        if w % 4 == 0:
            s = int(rng.integers(0, nsvc))
            gid = int(gids[s // sp][s % sp])
            for a, b in ((t - 100, t), (t - 4000, t - 20), (0, t)):  # (periods the 300-s ring or a longer level answers)
                assert eng.query_hist_period_stats(gid, a, b, t * 1_000_000, [25.0, 95.0]) == ring.period_stats(s, a, b, t, [25.0, 95.0])
        orc_win.window_clear(clear_hist=True)
        orc_all.window_clear(clear_hist=False)
    with pytest.raises(capi.GysError):
        eng.export_hist_level(0, t * 1_000_000, 0, nsvc)
    with pytest.raises(capi.GysError):
        eng.scan_listener_state(t * 1_000_000)
    eng.close()


# ------------------------------------------------------------------------------------------------ parity gaps named by VERDICT r2 (weak 3)
@pytest.mark.parametrize("td_cap", [0, 1920], ids=["cap896-library-default", "cap1920-bench-default"])
def test_c5_bench_shape_large_keys_over_several_pool_rounds(torch_mod, oracle, monkeypatch, td_cap):
    """the C5 bench shape (50 hosts x 2000 services, Zipf 1.1, one batch per window, the front end chosen by the engine = the split form)
    with the several-workgroup path of gys_huge.hpp walking its large keys in SEVERAL pool rounds (pool shrunk to 64 entries), two
    windows: every record, digest, buffer and register equals the oracle's bit for bit"""
    torch = torch_mod
    monkeypatch.setenv("GYS_HUGE_MAXENT", "64")
    nh, sp, n = 50, 2000, 1 << 24
    nsvc = nh * sp
    eng = _engine(max_hosts=nh, max_services=nsvc, max_batch_events=n, td_pend_cap=td_cap)
    orc = oracle.OracleEngine(nsvc, td_cap=td_cap)
    helpers.register_world(eng, orc, range(nh), sp)
    ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    for rnd in range(2):
        segs = eng.gen_resp_events(ev.data_ptr(), n, 0x5C5 + rnd, 0, nh, sp, 1100)
        eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
        eng.sync()
        orc.resp_batch(ev.cpu().numpy().tobytes(), [s.host_slot for s in segs], [s.first_event for s in segs])
        if rnd == 0:
            eng.window_close()
            orc.window_clear(clear_hist=True)
    c = eng.counters()
    assert c["resp_batches_host_split"] == 2 and c["resp_batches_general"] == 0
    per_batch = orc.hist()[:, 15, 0].astype(np.int64)
    assert (per_batch > 4096).sum() > 3 * 64, "the batch must carry several pool rounds of large keys"
    helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc.hist(), nsvc)
    assert (eng.export_conn_bitmap(0, nsvc) == orc.bitmap()).all()
    gs, gc, gm = eng.export_tdigest(0, nsvc)
    os_, oc, om = orc.td_arrays()
    assert (gc == oc).all() and (gs == os_).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, nsvc)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all() and (eng.export_cms(0) == orc.cms()).all()
    eng.close()


def test_c2_full_record_count(torch_mod, oracle):
    """BASELINE config 2 at its full 2^24 TCP_CONN_NOTIFY records per window (16 device-resident chunks of 2^20; 1 000 hosts x 100
    services): HLL and both Count-Min tables bit-exact vs the C oracle over all chunks; per-service connection counters == the number of
    CONNECTIONS the generator made (open + close notifications, accepting and connecting halves, loopback records: each connection once)"""
    import ctypes as C
    from gyeeta_amd import capi
    torch = torch_mod
    rng = np.random.default_rng(22)
    nh, sp, chunk, nchunks = 1000, 100, 1 << 20, 16
    eng = _engine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    s_ = np.arange(sp)
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "cluster%d" % (h % 8))
        eng.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))
    hll = np.zeros(1 << 14, dtype=np.uint8)
    cms32 = np.zeros(4 * 65536, dtype=np.uint32)
    cms64 = np.zeros(4 * 65536, dtype=np.uint64)
    nconn = np.zeros(nh * sp, dtype=np.int64)
    nclose = np.zeros(nh * sp, dtype=np.int64)
    tally = np.zeros(4, dtype=np.uint64)
    nconns_total = 0
    d_off = torch.arange(0, chunk * 280, 280, dtype=torch.int32, device="cuda")
    for k in range(nchunks):
        truth = {}
        rec = wire.synth_tcp_conns(rng, chunk, np.arange(nh), sp, dup_frac=0.2, v6_frac=0.05, truth=truth)
        raw = rec.tobytes()
        d_batch = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
        eng.order()
        capi.check(eng.L.gys_ingest_tcp_conn_dev(eng.h, C.c_void_p(d_batch.data_ptr()), C.c_void_p(d_off.data_ptr()), chunk))
        buf = np.frombuffer(raw, dtype=np.uint8)
        assert oracle.lib().gyo_tcp_conn_sketch_batch(buf.ctypes.data, chunk, buf.ctypes.data + len(buf), oracle.ptr(hll, oracle.u8p),
                                                       oracle.ptr(cms32, oracle.u32p), oracle.ptr(cms64, oracle.u64p)) == chunk
        assert oracle.lib().gyo_tcp_conn_walk_tallies(buf.ctypes.data, chunk, buf.ctypes.data + len(buf), oracle.ptr(tally, oracle.u64p)) == chunk
        # ground truth = the generator's connection table (registration order: slot = h * sp + s), not the records' flag bytes
        slot = truth["conn_host"].astype(np.int64) * sp + truth["conn_svc"]
        nconn += np.bincount(slot, minlength=nh * sp)
        nclose += np.bincount(slot[truth["conn_closed"]], minlength=nh * sp)
        nconns_total += len(slot)
        eng.sync()  # (the chunk's device buffer is released by torch once it goes out of scope)
    eng.window_close()
    assert (eng.export_hll() == hll).all()
    assert (eng.export_cms(0).ravel() == cms32).all()
    assert (eng.export_cms(1).ravel().astype(np.uint64) == cms64).all()
    ctr = eng.export_svc_counters()
    assert (ctr[:, 0].astype(np.int64) == nconn).all() and int(ctr[:, 0].sum()) == nconns_total < chunk * nchunks
    assert (ctr[:, 1].astype(np.int64) == nclose).all()
    c = eng.counters()
    assert c["conn_events"] == chunk * nchunks
    assert [c["conn_new"], c["conn_closed"], c["conn_closed_no_notify"], c["conn_client_side"]] == tally.tolist()
    eng.close()
