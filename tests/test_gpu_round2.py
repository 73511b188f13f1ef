"""Round-2 parity additions (VERDICT r1 "what's weak" 2-4):
  * per-service distinct-client HLL (svc_hll_p) against an oracle built from the reference's own flow-key bytes;
  * the all-time view mid-window in both record modes (lazily folded records / per-event records);
  * re-registration of a known glob_id keeps the slot and its state;
  * the window exchange driven through the ENGINE with nranks = 2 (two processes on one GPU, gloo all-reduce of the four register
    sections): reduced registers, Count-Min, all-service histogram and gys_query_clusterstate equal the single-rank run."""
import os
import socket

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


def _flow_words(ev):
    """the bytes PAIR_IP_PORT(cli = daddr:dport, ser = saddr:sport).get_hash() hashes (common/gy_inet_inc.h:225-247) as u32 words:
    an IPv4 address is 1 word when ip32_be != 0, else the 16 zero bytes of ip128_be (GY_IP_ADDR::get_as_inaddr)"""
    out = []
    for daddr, dport, saddr, sport in zip(ev["daddr"].tolist(), ev["dport_be"].tolist(), ev["saddr"].tolist(), ev["sport_be"].tolist()):
        w = ([daddr] if daddr else [0, 0, 0, 0]) + [dport] + ([saddr] if saddr else [0, 0, 0, 0]) + [sport]
        out.append(np.array(w, dtype=np.uint32))
    return out


@pytest.mark.parametrize("resp_path", [1, 2], ids=["general", "hostlocal"])
def test_svc_hll_registers_bit_exact(torch_mod, oracle, resp_path):
    rng = np.random.default_rng(21)
    nh, sp, P = 2, 5, 8
    eng = _engine(max_hosts=4, max_services=32, max_batch_events=1 << 16, svc_hll_p=P, resp_path=resp_path)
    orc = oracle.OracleEngine(32)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    regs = np.zeros((nh * sp, 1 << P), dtype=np.uint8)
    L = oracle.lib()
    for rnd in range(3):
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, 2500, sp, zero_ip_frac=0.05)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
            lat = (ev["lsndtime"] - ev["lrcvtime"]).astype(np.uint32)
            svc = ev["sport_be"].astype(np.int64) - 1024
            keep = (lat <= 1000000) & (svc >= 0) & (svc < sp)
            for w, s, k in zip(_flow_words(ev), svc.tolist(), keep.tolist()):
                if k:
                    slot = eng.lookup(int(gids[h][s]))
                    L.gyo_hll_add_words(oracle.ptr(regs[slot], oracle.u8p), P, oracle.ptr(w, oracle.u32p), len(w))
    eng.sync()
    got = eng.export_svc_hll(0, nh * sp)
    assert (got == regs).all(), f"per-service HLL registers differ at {np.argwhere(got != regs)[:4].tolist()}"
    eng.window_close()
    assert (eng.export_hll() == orc.hll()).all()  # the global registers (of the closed window) come out of the same hash
    assert eng.export_svc_hll(0, nh * sp).sum() == 0  # per-window registers
    eng.close()


@pytest.mark.parametrize("td", [True, False], ids=["lazy_records", "per_event_records"])
def test_alltime_view_includes_open_window_in_both_modes(torch_mod, oracle, td):
    """which = 1 answers "everything ingested so far" mid-window whichever record mode runs (gysketch.h)"""
    rng = np.random.default_rng(23)
    eng = _engine(max_hosts=2, max_services=8, max_batch_events=1 << 14, enable_tdigest=td)
    cum = oracle.OracleEngine(8, enable_td=False)   # never cleared: the all-time truth
    win = oracle.OracleEngine(8, enable_td=False)   # cleared at every boundary: the open window
    info, gids = helpers.register_world(eng, cum, range(1), 3)
    helpers.register_world(None, win, range(1), 3)
    for w in range(3):
        for b in range(2):
            ev = helpers.make_resp_events(rng, 0, 1500 + 400 * b, 3)
            eng.handle_resp_events(info[0][0], ev)
            cum.resp_batch(ev.tobytes(), [0], [0])
            win.resp_batch(ev.tobytes(), [0], [0])
            # mid-window: window view = this window's events, all-time view = every event so far
            helpers.assert_hist_equal(eng.export_hist(0, 0, 3), win.hist(), 3)
            helpers.assert_hist_equal(eng.export_hist(1, 0, 3), cum.hist(), 3)
            g = int(gids[0][1])
            vals, sums, counts, tot, mx, avg = eng.hist_percentiles(g, [50.0, 99.0], which=1)
            h = cum.hist()[1]
            ov, os_, ocn, oavg = oracle.hist_percentiles(0, h[:15], h[15][0], [50.0, 99.0])
            assert vals == ov and counts == ocn and tot == h[15][0]
        eng.window_close()
        win.window_clear(clear_hist=True)
        helpers.assert_hist_equal(eng.export_hist(1, 0, 3), cum.hist(), 3)
    eng.close()


def test_reregistration_keeps_slot_and_state(torch_mod, oracle):
    rng = np.random.default_rng(24)
    eng = _engine(max_hosts=2, max_services=16, max_batch_events=1 << 14)
    orc = oracle.OracleEngine(16)
    info, gids = helpers.register_world(eng, orc, range(1), 4)
    mid = info[0][0]
    ev = helpers.make_resp_events(rng, 0, 3000, 4)
    eng.handle_resp_events(mid, ev)
    orc.resp_batch(ev.tobytes(), [0], [0])
    before = eng.num_services()
    s = np.arange(6)  # the partha reconnects and resends its 4 listeners plus two new ones, one of them twice
    g = wire.glob_id(np.full(6, 0), s)
    g[5] = g[4]
    first = eng.register_listeners_np(mid, g, wire.listener_netns(0, s), wire.listener_port(s))
    assert first == before and eng.num_services() == before + 1
    for i in range(4):
        assert eng.lookup(int(gids[0][i])) == i
    orc.register(0, int(g[4]), int(wire.listener_netns(0, s)[4]), int(wire.listener_port(s)[4]))
    ev = helpers.make_resp_events(rng, 0, 3000, 5)
    eng.handle_resp_events(mid, ev)
    orc.resp_batch(ev.tobytes(), [0], [0])
    eng.sync()
    helpers.assert_hist_equal(eng.export_hist(0, 0, 5), orc.hist(), 5)
    gs, gc, gm = eng.export_tdigest(0, 5)
    os_, oc, om = orc.td_arrays()
    assert (gs == os_[:5]).all() and (gc == oc[:5]).all()
    eng.close()


# ------------------------------------------------------------------------------------------------ nranks = 2 through the engine
NH, SP, NEV = 10, 6, 4000
CLUSTERS = ["east", "west", "north"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _feed(eng, hosts, rank_of=None):
    """registers `hosts` and feeds every ingest path; identical bytes whatever the sharding (per-host seeds)"""
    for c in CLUSTERS:  # same cluster order on every rank (gysketch.h)
        eng.register_cluster(c)
    for h in hosts:
        mid = wire.machine_id(h)
        eng.register_host(mid, CLUSTERS[h % 3])
        s = np.arange(SP)
        eng.register_listeners_np(mid, wire.glob_id(np.full(SP, h), s), wire.listener_netns(h, s), wire.listener_port(s))
    for h in hosts:
        rng = np.random.default_rng(1000 + h)
        mid = wire.machine_id(h)
        eng.handle_resp_events(mid, helpers.make_resp_events(rng, h, NEV, SP))
        rec = wire.synth_tcp_conns(rng, 300, [h], SP, dup_frac=0.1)
        eng.partha_tcp_conn_info(mid, wire.pack_variable(rec, [b""] * 300), 300)
        ls = wire.synth_listener_states(rng, h, np.arange(SP))
        eng.partha_listener_state(mid, wire.pack_variable(ls, [b""] * len(ls)), len(ls))
        eng.handle_host_state(mid, ntasks=50 + h, nlisten=SP, ntasks_issue=h % 2)


def _observe(eng):
    gh = eng.export_global_hist()
    cl = tuple(eng.clusterstate(c).as_tuple() for c in CLUSTERS)
    return (eng.export_hll().tobytes(), eng.export_cms(0).tobytes(), eng.export_cms(1).tobytes(),
            tuple((gh.stats[i].count, gh.stats[i].sum) for i in range(15)) + ((gh.total_count, gh.max_val_seen),), cl,
            round(eng.distinct_flows(), 6))


def _rank_worker(rank, port, q):
    import torch
    import torch.distributed as dist
    from gyeeta_amd import capi
    from gyeeta_amd.engine import SketchEngine, mid_buf
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        L = capi.load()
        mine = [h for h in range(NH) if L.gys_shard_of(mid_buf(wire.machine_id(h)), 2) == rank]
        other = [h for h in range(NH) if h not in mine]
        eng = SketchEngine(max_hosts=NH, max_services=NH * SP, max_clusters=4, max_batch_events=1 << 14, rank=rank, nranks=2, device=0)
        _feed(eng, mine)
        not_owner = False
        try:  # a host of the other shard is refused (GYS_ERR_NOT_OWNER): its partha belongs to the other madhava
            eng.register_host(wire.machine_id(other[0]), CLUSTERS[0])
        except capi.GysError as e:
            not_owner = e.code == capi.ERR_NOT_OWNER
        eng.window_close(tusec=5_000_000)   # gys_window_prepare -> all-reduce of the 4 sections (gloo) -> gys_window_finish
        obs = _observe(eng)
        eng.close()
        q.put((rank, obs, len(mine), not_owner))
    finally:
        dist.destroy_process_group()


def test_window_exchange_through_engine_world2(torch_mod):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _engine(max_hosts=NH, max_services=NH * SP, max_clusters=4, max_batch_events=1 << 14)
    _feed(single, range(NH))
    single.window_close(tusec=5_000_000)
    want = _observe(single)
    single.close()
    assert res[0][2] > 0 and res[1][2] > 0 and res[0][2] + res[1][2] == NH
    assert res[0][3] and res[1][3], "registering a host of the other shard must fail with GYS_ERR_NOT_OWNER"
    names = ["hll", "cms32", "cms64", "global histogram", "cluster state", "distinct flows"]
    for r in range(2):
        for name, got, exp in zip(names, res[r][1], want):
            assert got == exp, f"rank {r}: {name} differs from the single-rank run"


def test_scan_quantiles_equals_per_key_queries_and_oracle(torch_mod, oracle):
    """the per-key scan on the digests: ONE device pass gives the quantiles of every service -- keys with merged
    clusters + buffered values, buffer-only keys, freshly merged keys (empty buffer) and keys without events -- identical to the
    one-key query and to the oracle's merged view, and nothing is modified"""
    import ctypes as C
    torch = torch_mod
    nh, sp, n = 32, 50, 1 << 19
    eng = _engine(max_hosts=nh + 1, max_services=nh * sp + 8, max_batch_events=n)
    orc = oracle.OracleEngine(nh * sp + 8)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    for rnd in range(4):  # ~330 values per key and round: most keys re-cluster in round 3, some in round 4
        segs = eng.gen_resp_events(ev.data_ptr(), n, 777 + rnd, 0, nh, sp)
        eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
        eng.sync()
        orc.resp_batch(ev.cpu().numpy().tobytes(), [s.host_slot for s in segs], [s.first_event for s in segs])
    rng = np.random.default_rng(3)
    small = helpers.make_resp_events(rng, 5, 40, 3, bad_frac=0.0, unknown_frac=0.0)   # a few more values for three keys of host 5
    eng.handle_resp_events(info[5][0], small)
    orc.resp_batch(small.tobytes(), [info[5][1]], [0])
    extra = wire.machine_id(40)  # a host with two buffer-only services (60 events) and two that never see an event
    xslot = eng.register_host(extra, "cluster0")
    s = np.arange(4)
    eng.register_listeners_np(extra, wire.glob_id(np.full(4, 40), s), wire.listener_netns(40, s), wire.listener_port(s))
    for i in range(4):
        orc.register(xslot, int(wire.glob_id(40, i)), int(wire.listener_netns(40, s)[i]), int(wire.listener_port(s)[i]))
    few = helpers.make_resp_events(rng, 40, 60, 2, bad_frac=0.0, unknown_frac=0.0)
    eng.handle_resp_events(extra, few)
    orc.resp_batch(few.tobytes(), [xslot], [0])
    nsvc = eng.num_services()
    qs = [0.0, 0.25, 0.5, 0.95, 0.99, 1.0]
    before = (eng.export_tdigest(0, nsvc), eng.export_tdigest_pending(0, nsvc))
    got = eng.scan_quantiles(qs)
    assert got.shape == (nsvc, len(qs))
    gn, _ = before[1]
    gs, gc, gm = before[0]
    assert (gn > 0).any() and (gn == 0).any() and (gc.sum(axis=1) > 0).any() and ((gc.sum(axis=1) == 0) & (gn > 0)).any()
    L = oracle.lib()
    for slot in range(nh * sp + 2):
        want = [L.gyo_tdb_quantile(C.byref(orc.td(slot)), q) for q in qs]
        assert got[slot].tolist() == want, (slot, got[slot].tolist(), want)
    assert (got[nh * sp + 2:] == 0.0).all()  # no events: 0, as gys_query_quantiles answers
    for h, sv in ((0, 0), (5, 1), (31, 49)):
        g = int(gids[h][sv])
        assert eng.quantiles(g, qs) == got[eng.lookup(g)].tolist()
    after = (eng.export_tdigest(0, nsvc), eng.export_tdigest_pending(0, nsvc))
    for x, y in zip(before[0] + before[1], after[0] + after[1]):
        assert (x == y).all()
    eng.close()


def test_tdigest_rollup_host_cluster_global_bit_exact(torch_mod, oracle):
    """merged digests of groups of services: per host, per cluster and over all hosts, every slab equal to the
    oracle's roll-up (oracle/gy_oracle_rollup.c: the union by value bin) bit for bit -- members with clusters + buffered values, buffer-only members,
    empty members; then the cross-rank form: slabs of two engines (two ranks' shards, different streams), concatenated as an all-gather would
    and rolled up, equal to the oracle's roll-up of the two global slabs"""
    import ctypes as C
    torch = torch_mod
    from gyeeta_amd import capi
    L = oracle.lib()
    qs = [0.01, 0.25, 0.5, 0.95, 0.99, 0.999]

    def world(hosts, seed):
        nh, sp, n = len(hosts), 40, 1 << 18
        eng = _engine(max_hosts=nh + 1, max_services=(nh + 1) * sp, max_batch_events=n, max_clusters=4)
        orc = oracle.OracleEngine((nh + 1) * sp)
        for cname in ("cluster0", "cluster1", "cluster2"):
            eng.register_cluster(cname)
        info, gids = helpers.register_world(eng, orc, hosts, sp)
        ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        for rnd in range(5):  # ~330 values per key and round over the first nh hosts: most keys re-cluster once, all keep a buffer
            segs = eng.gen_resp_events(ev.data_ptr(), n, seed + rnd, 0, nh, sp)
            eng.handle_resp_events_dev(segs, ev.data_ptr(), n)
            eng.sync()
            orc.resp_batch(ev.cpu().numpy().tobytes(), [s.host_slot for s in segs], [s.first_event for s in segs])
        quiet = wire.machine_id(900 + seed)  # a host without events, and one with buffer-only services
        eng.register_host(quiet, "cluster1")
        s = np.arange(3)
        eng.register_listeners_np(quiet, wire.glob_id(np.full(3, 900 + seed), s), wire.listener_netns(900 + seed, s), wire.listener_port(s))
        for i in range(3):
            orc.register(nh, int(wire.glob_id(900 + seed, i)), int(wire.listener_netns(900 + seed, s)[i]), int(wire.listener_port(s)[i]))
        return eng, orc, nh, sp

    def oracle_host_slabs(orc, nh, sp):
        return [oracle.rollup_services([orc.td(h * sp + k) for k in range(sp if h < nh else 3)]) for h in range(nh + 1)]

    fold = oracle.rollup_slabs  # (round 6: the roll-up is the union by value bin, oracle/gy_oracle_rollup.c gyo_tdbins_*)

    def same(rec, d):
        ok = (rec["sum"] == np.array(d.sum[:], dtype=np.int64)).all() and (rec["cnt"] == np.array(d.cnt[:], dtype=np.uint64)).all()
        if L.gyo_td64_total(C.byref(d)):
            ok = ok and int(rec["vmin"]) == d.vmin and int(rec["vmax"]) == d.vmax
        return bool(ok)

    hosts_a, hosts_b = list(range(6)), list(range(5))  # (the device generator draws for host indices 0..n-1: two engines = two ranks' shards)
    globals_dev, globals_orc = [], []
    for hosts, seed in ((hosts_a, 31), (hosts_b, 57)):
        eng, orc, nh, sp = world(hosts, seed)
        ohs = oracle_host_slabs(orc, nh, sp)
        before = eng.export_tdigest(0, eng.num_services())
        dev_h, rec_h = eng.tdigest_rollup(capi.ROLLUP_HOST)
        assert len(rec_h) == nh + 1
        for h in range(nh + 1):
            assert same(rec_h[h], ohs[h]), f"host slab {h} differs"
        assert rec_h[nh]["cnt"].sum() == 0  # the quiet host: an empty digest
        assert eng.slab_quantiles(dev_h, qs, index=2) == [L.gyo_td64_quantile(C.byref(ohs[2]), q) for q in qs]
        dev_c, rec_c = eng.tdigest_rollup(capi.ROLLUP_CLUSTER)
        cl_of = [hosts[h] % 3 for h in range(nh)] + [1]  # helpers.register_world: cluster%d of (host index % 3); the quiet host: cluster1
        for cl in range(3):
            assert same(rec_c[cl], fold([ohs[h] for h in range(nh + 1) if cl_of[h] == cl])), f"cluster slab {cl} differs"
        dev_g, rec_g = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
        og = fold(ohs)
        assert same(rec_g[0], og)
        assert int(rec_g[0]["cnt"].sum()) == L.gyo_td64_total(C.byref(og)) > 0
        assert eng.slab_quantiles(dev_g, qs) == [L.gyo_td64_quantile(C.byref(og), q) for q in qs]
        after = eng.export_tdigest(0, eng.num_services())
        assert all((x == y).all() for x, y in zip(before, after))  # nothing modified
        globals_dev.append(dev_g.clone())
        globals_orc.append(og)
        if seed == 57:  # cross-rank merge on this engine: the two ranks' global slabs as an all-gather lays them out
            gathered = torch.cat(globals_dev)
            dev_m, rec_m = eng.tdigest_merge_slabs(gathered, 2)
            om = fold(globals_orc)
            assert same(rec_m, om)
            assert eng.slab_quantiles(dev_m, qs) == [L.gyo_td64_quantile(C.byref(om), q) for q in qs]
        eng.close()


def _rccl_worker(q):
    """body of test_window_close_rccl_inside_the_library, in its own process: RCCL's bootstrap (ncclCommInitRank) does not return on part
    of the GPU pool, and a hang inside a C call cannot be interrupted from within the process"""
    import ctypes as C
    import torch
    from gyeeta_amd import capi
    from gyeeta_amd.engine import SketchEngine
    from oracle import oracle
    try:
        rng = np.random.default_rng(31)
        engs = [SketchEngine(max_hosts=4, max_services=32, max_batch_events=1 << 14) for _ in range(2)]
        for e in engs:
            for h in range(3):
                mid = wire.machine_id(h)
                e.register_host(mid, "cluster%d" % (h % 2))
                s = np.arange(5)
                e.register_listeners_np(mid, wire.glob_id(np.full(5, h), s), wire.listener_netns(h, s), wire.listener_port(s))
        L = engs[0].L
        uid = (C.c_uint8 * 128)()
        capi.check(L.gys_rccl_unique_id(uid))
        comm = C.c_void_p()
        q.put("joining")
        rc = L.gys_rccl_comm_create(engs[0].h, uid, 1, 0, C.byref(comm))
        if rc != capi.OK:  # RCCL's own bootstrap failed on this box (it also hangs on part of the pool): nothing of the library to test
            q.put("bootstrap-failed: " + L.gys_last_error().decode(errors="replace"))
            return
        q.put("joined")
        for w in range(2):
            for h in range(3):
                ev = helpers.make_resp_events(rng, h, 3000, 5)
                rec = wire.synth_tcp_conns(rng, 200, [h], 5)
                ls = wire.synth_listener_states(rng, h, np.arange(5))
                for e in engs:
                    e.handle_resp_events(wire.machine_id(h), ev)
                    e.partha_tcp_conn_info(wire.machine_id(h), wire.pack_variable(rec, [b""] * 200), 200)
                    e.partha_listener_state(wire.machine_id(h), wire.pack_variable(ls, [b""] * 5), 5)
                    e.handle_host_state(wire.machine_id(h), ntasks=10 + h, nlisten=5)
            capi.check(L.gys_window_close_rccl(engs[0].h, comm, 5_000_000 * (w + 1)))
            engs[1].window_close(tusec=5_000_000 * (w + 1))
            a, b = engs
            assert (a.export_hll() == b.export_hll()).all() and (a.export_cms(0) == b.export_cms(0)).all() and (a.export_cms(1) == b.export_cms(1)).all()
            assert a.clusterstate("cluster0").as_tuple() == b.clusterstate("cluster0").as_tuple()
            ga, gb = a.export_global_hist(), b.export_global_hist()
            assert ga.total_count == gb.total_count > 0 and ga.max_val_seen == gb.max_val_seen
        out = torch.zeros(C.sizeof(capi.TDigestSlab), dtype=torch.uint8, device="cuda")
        capi.check(L.gys_tdigest_global_rccl(engs[0].h, comm, C.c_void_p(out.data_ptr())))
        engs[0].sync()
        got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=engs[0].SLAB_DT)[0]
        _, loc = engs[0].tdigest_rollup(capi.ROLLUP_GLOBAL)
        o1 = oracle.TD64()
        o1.sum[:] = loc[0]["sum"].tolist()
        o1.cnt[:] = loc[0]["cnt"].tolist()
        o1.vmin, o1.vmax = int(loc[0]["vmin"]), int(loc[0]["vmax"])
        d = oracle.rollup_slabs([o1])
        assert got["cnt"].sum() == loc[0]["cnt"].sum() > 0
        assert (got["sum"] == np.array(d.sum[:], dtype=np.int64)).all() and (got["cnt"] == np.array(d.cnt[:], dtype=np.uint64)).all()
        capi.check(L.gys_rccl_comm_destroy(comm))
        for e in engs:
            e.close()
        q.put("ok")
    except BaseException as ex:  # noqa: BLE001 -- reported to the parent
        import traceback
        q.put("error: " + "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-1500:])


def test_window_close_rccl_inside_the_library(torch_mod):
    """gys_window_close_rccl / gys_tdigest_global_rccl with a one-rank communicator created through the C ABI (the box has one GPU):
    the registers after the in-library exchange equal those of a twin engine closed without it, and the all-gathered + rolled-up global
    digest equals the oracle's roll-up of the local GYS_ROLLUP_GLOBAL slab"""
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    seen = []
    try:
        while True:
            msg = q.get(timeout=75 if seen else 150)  # (the first message comes after the worker's own import of torch)
            seen.append(msg)
            if msg == "ok" or msg.startswith("error") or msg.startswith("bootstrap-failed"):
                break
    except queue.Empty:
        p.kill()
        p.join(timeout=30)
        if seen and seen[-1] in ("joining", "joined"):
            pytest.skip(f"RCCL did not return within 75 s on this box (after {seen[-1]!r}); the in-library exchange was not exercised")
        pytest.fail(f"RCCL window worker stalled after {seen}")
    p.join(timeout=60)
    if seen[-1].startswith("bootstrap-failed"):
        pytest.skip("ncclCommInitRank returned an error on this box (" + seen[-1] + "); the in-library exchange was not exercised")
    assert seen[-1] == "ok", seen[-1]
