"""Round 5 GPU parity (SURVEY 8 row a7 finished): IPv6 response events (handle_ipv6_resp_event, common/gy_socket_stat.cc:1535-1551), listeners
bound to an address / several listeners on one (netns, port) (operator==(shared_ptr<TCP_LISTENER>, NS_IP_PORT), common/gy_socket_stat.h:708-714;
insert_or_replace common/gy_socket_stat.cc:1372), and LISTENER_STATE_NOTIFY records that name a listener more than once handed to the DEVICE
entry point -- each compared with the oracle's serial walk through the C-ABI library."""
import ctypes as C
import ipaddress

import numpy as np
import pytest

from gyeeta_amd import capi, wire
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


def v4(s):
    return ipaddress.IPv4Address(s).packed


def v6(s):
    return ipaddress.IPv6Address(s).packed


def mapped(b4):
    return bytes(10) + b"\xff\xff" + bytes(b4)


class World:
    """hosts with `nplain` listeners that are alone on their port and any-address (ports 1024 ..) and, per host, the special keys:
         port 80: bound A, then bound B6 (IPv6), then any-address          (first match in registration order)
         port 81: any-address, then bound B                                 (the bound one REPLACES the any-address one in place)
         port 82: bound B, alone                                            (events for other addresses find nobody)
         port 83: bound ::ffff:A (IPv6 form of an IPv4 address)             (equals A: ip32_be_ shares storage with embedded_ipv4_)
         port 84: bound 2002:<C>::1 (6to4 of C), then bound C               (the second REPLACES the first: the addresses compare equal)"""

    def __init__(self, eng, orc, hosts, nplain, netns_of=lambda h: 0xF0000000 + 4 * h):
        self.info, self.addr = {}, {}
        self.netns_of = netns_of
        self.nplain = nplain
        for h in hosts:
            mid = wire.machine_id(h)
            slot = eng.register_host(mid, "cluster%d" % (h % 3))
            A, B, Cc = bytes([10, 1, h & 255, 3]), bytes([10, 2, h & 255, 5]), bytes([10, 3, h & 255, 7])
            B6 = v6("2001:db8::%x:6" % (h + 1))
            C6 = b"\x20\x02" + Cc + bytes(9) + b"\x01"
            self.addr[h] = dict(A=A, B=B, C=Cc, B6=B6, C6=C6)
            ns = netns_of(h)
            gid = [0x7000000 * (h + 1)]

            def reg(port, addr, eng=eng, orc=orc, mid=mid, slot=slot, ns=ns, gid=gid):
                gid[0] += 1
                s_e = eng.register_listeners(mid, [gid[0]], [ns], [port], addrs=[addr])
                s_o = orc.register_addr(slot, gid[0], ns, port, addr, is_v6=addr is not None and len(addr) == 16)
                assert s_e == s_o
                return s_e

            sl = {}
            # plain listeners first, in one call
            g = np.arange(nplain, dtype=np.uint64) + np.uint64(gid[0] + 1000)
            ports = 1024 + np.arange(nplain)
            first = eng.register_listeners_np(mid, g, np.full(nplain, ns), ports)
            for i in range(nplain):
                assert orc.register(slot, int(g[i]), ns, int(ports[i])) == first + i
            sl["80A"], sl["80B6"], sl["80any"] = reg(80, A), reg(80, B6), reg(80, None)
            sl["81old"], sl["81B"] = reg(81, None), reg(81, B)
            sl["82B"] = reg(82, B)
            sl["83mA"] = reg(83, mapped(A))
            sl["84C6"], sl["84C"] = reg(84, C6), reg(84, Cc)
            self.info[h] = (mid, slot, sl)

    def ports(self, rng, n):
        """a port per event: the plain ones, the special keys, and one nobody listens on"""
        special = np.array([80, 81, 82, 83, 84, 999])
        return np.where(rng.random(n) < 0.5, special[rng.integers(0, len(special), n)], 1024 + rng.integers(0, self.nplain, n))

    def events4(self, rng, h, n):
        a = self.addr[h]
        ev = np.zeros(n, dtype=wire.RESP_EVENT)
        pool = np.array([int.from_bytes(x, "little") for x in (a["A"], a["B"], a["C"], bytes([10, 9, 9, 9]), bytes(4))], dtype=np.uint32)
        ev["saddr"] = pool[rng.integers(0, len(pool), n)]
        ev["daddr"] = np.where(rng.random(n) < 0.02, 0, (0x0B000000 | rng.integers(0, 1 << 14, n)).astype(">u4").view("<u4"))
        ev["netns"] = self.netns_of(h)
        ev["sport_be"] = self.ports(rng, n)
        ev["dport_be"] = rng.integers(16000, 65536, n)
        self._times(rng, ev, n)
        return ev

    def events6(self, rng, h, n):
        a = self.addr[h]
        ev = np.zeros(n, dtype=wire.RESP_EVENT6)
        spool = np.frombuffer(b"".join([a["B6"], mapped(a["A"]), mapped(a["B"]), a["C6"], mapped(a["C"]), v6("2001:db8::dead"), bytes(16), v6("::1"),
                                        b"\x00\x64\xff\x9b" + bytes(8) + a["B"]]), dtype=np.uint8).reshape(-1, 16)
        ev["saddr"] = spool[rng.integers(0, len(spool), n)]
        d = np.zeros((n, 16), dtype=np.uint8)
        kind = rng.integers(0, 6, n)
        low = rng.integers(0, 1 << 13, n)
        d[:, 14], d[:, 15] = low >> 8, low & 255
        d[kind == 0, 10:12] = 0xFF                                    # ::ffff:0.0.x.y style mapped clients (embedded address)
        d[kind == 1, 0], d[kind == 1, 1] = 0x20, 0x02                  # 6to4 clients: the embedded address is bytes 2..5
        d[kind == 1, 4] = (low >> 8)[kind == 1]
        d[kind == 2, 1], d[kind == 2, 2], d[kind == 2, 3] = 0x64, 0xFF, 0x9B  # NAT64 clients
        d[kind == 3, 0] = 0xFD                                         # unique-local clients
        d[kind == 4, 0], d[kind == 4, 1] = 0x20, 0x01                  # global clients
        d[kind == 5] = 0                                               # :: (hashes as 16 zero bytes)
        ev["daddr"] = d
        ev["netns"] = self.netns_of(h)
        ev["sport_be"] = self.ports(rng, n)
        ev["dport_be"] = rng.integers(16000, 65536, n)
        self._times(rng, ev, n)
        return ev

    @staticmethod
    def _times(rng, ev, n):
        lat = np.minimum(np.floor(rng.lognormal(3.0, 1.6, n)), 1e6).astype(np.uint32)
        bad = rng.random(n) < 0.02
        lat = np.where(bad, np.uint32(1000001), lat)
        lrcv = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        with np.errstate(over="ignore"):
            ev["lsndtime"] = lrcv + lat
        ev["lrcvtime"] = lrcv


def _compare(eng, orc, check_window=True):
    n = orc.nsvc
    helpers.assert_hist_equal(eng.export_hist(0, 0, n), orc.hist(), n)
    gb, ob = eng.export_conn_bitmap(0, n), orc.bitmap()
    assert gb.shape == ob.shape == (n, 64)
    assert (gb == ob).all(), f"CONN_BITMAP rows differ at {np.argwhere(gb != ob)[:4].tolist()}"
    gs, gc, gm = eng.export_tdigest(0, n)
    os_, oc, om = orc.td_arrays()
    assert (gc == oc).all() and (gs == os_).all() and (gm == om).all()
    gn, gp = eng.export_tdigest_pending(0, n)
    on, op = orc.td_pending()
    assert (gn == on).all() and (gp == op).all()
    c, oc4 = eng.counters(), orc.counters()
    assert [c["resp_events"], c["resp_dropped_range"], c["resp_dropped_nolistener"]] == [oc4["events"], oc4["dropped_range"], oc4["dropped_nolistener"]]


def _compare_window(eng, orc):
    assert (eng.export_hll() == orc.hll()).all()
    assert (eng.export_cms(0) == orc.cms()).all()
    gh = eng.export_global_hist()
    oh, omax = orc.ghist()
    assert [gh.stats[i].count for i in range(15)] == oh[:15, 0].tolist() and [gh.stats[i].sum for i in range(15)] == oh[:15, 1].tolist()
    assert gh.total_count == oh[15, 0] and gh.max_val_seen == omax


@pytest.mark.parametrize("resp_path", [1, 2], ids=["general", "hostlocal"])
def test_mixed_v4_v6_stream_with_bound_listeners(torch_mod, oracle, resp_path):
    """a replayed agent stream with both families and listeners bound to addresses: per-listener histogram, CONN_BITMAP rows of both families,
    digest, buffered values, HLL / Count-Min / all-service histogram and the three drop counters equal the oracle's serial walk"""
    rng = np.random.default_rng(51)
    nh, nplain = 4, 40
    eng = _engine(max_hosts=8, max_services=1024, max_batch_events=1 << 18, resp_path=resp_path)
    orc = oracle.OracleEngine(1024)
    w = World(eng, orc, range(nh), nplain)
    for rnd in range(5):
        for h in range(nh):
            mid, slot, _ = w.info[h]
            for fam in (rng.permutation(2) if rnd else (0, 1)):
                n = int(rng.integers(1, 30000 if rnd == 3 else 5000))
                if fam == 0:
                    ev = w.events4(rng, h, n)
                    eng.handle_resp_events(mid, ev)
                    orc.resp_batch(ev.tobytes(), [slot], [0])
                else:
                    ev = w.events6(rng, h, n)
                    eng.handle_resp_events_v6(mid, ev)
                    orc.resp_batch_v6(ev.tobytes(), [slot], [0])
        eng.handle_resp_events_v6(w.info[0][0], np.zeros(0, dtype=wire.RESP_EVENT6))  # empty batch is a no-op
        eng.sync()
        _compare(eng, orc)
    # the rule itself, on the slots: the replaced any-address listener of port 81 got nothing, the IPv6-form listener of port 83 got IPv4 events
    hist = eng.export_hist(0, 0, orc.nsvc)
    bm = eng.export_conn_bitmap(0, orc.nsvc)
    for h in range(nh):
        sl = w.info[h][2]
        assert hist[sl["81old"]][15][0] == 0 and hist[sl["81B"]][15][0] > 0
        assert hist[sl["84C6"]][15][0] == 0 and hist[sl["84C"]][15][0] > 0
        assert bm[sl["83mA"], :32].any() and bm[sl["83mA"], 32:].any()      # A as IPv4 events and as ::ffff:A IPv6 events
        assert bm[sl["80B6"], 32:].any() and not bm[sl["80B6"], :32].any()  # an IPv6-only address never sees an IPv4 event
    c = eng.counters()
    assert c["resp_dropped_nolistener"] > 0
    if resp_path == 1:
        assert c["resp_batches_host_local"] == 0
    else:
        assert c["resp_batches_general"] == 0
    eng.window_close()
    _compare_window(eng, orc)
    eng.close()


def test_device_batches_of_both_families_many_hosts_and_parts(torch_mod, oracle):
    """device-resident multi-host batches (gys_ingest_resp_events_dev / _v6_dev) incl. a host with more listeners than one LDS sub-table
    takes (cut into parts: every part's workgroup resolves candidates through the host's one region) and window boundaries in between"""
    torch = torch_mod
    rng = np.random.default_rng(52)
    nh = 6
    eng = _engine(max_hosts=8, max_services=8192, max_batch_events=1 << 20)
    orc = oracle.OracleEngine(8192)
    worlds = [World(eng, orc, [h], 2600 if h == 2 else 150) for h in range(nh)]  # host 2: 2609 listeners -> two parts
    info = {h: worlds[h].info[h] for h in range(nh)}
    for rnd in range(4):
        for fam in (0, 1):
            parts, segs_h, first = [], [], 0
            segs = (capi.RespSeg * nh)()
            for i, h in enumerate(rng.permutation(nh)):
                n = int(rng.integers(2000, 60000 if h == 2 else 20000))
                ev = worlds[h].events4(rng, h, n) if fam == 0 else worlds[h].events6(rng, h, n)
                parts.append(ev.tobytes())
                segs[i].host_slot, segs[i].first_event = info[h][1], first
                segs_h.append((info[h][1], first))
                first += n
            raw = b"".join(parts)
            d = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
            if fam == 0:
                eng.handle_resp_events_dev(segs, d.data_ptr(), first)
                orc.resp_batch(raw, [s for s, _ in segs_h], [f for _, f in segs_h])
            else:
                eng.handle_resp_events_v6_dev(segs, d.data_ptr(), first)
                orc.resp_batch_v6(raw, [s for s, _ in segs_h], [f for _, f in segs_h])
            eng.sync()
        _compare(eng, orc)
        if rnd == 1:
            eng.window_close()
            _compare_window(eng, orc)
            orc.window_clear(clear_hist=True)  # (_compare looks at the window view of the records)
    c = eng.counters()
    assert c["resp_batches_general"] == 0 and c["resp_batches_host_local"] > 0
    eng.close()


def test_listener_reregistration_follows_insert_or_replace(torch_mod, oracle):
    """registration order decides: a later listener takes the place of the first earlier one it compares equal to (or that is any-address);
    a reconnecting partha that resends a glob_id keeps the slot.  Engine and oracle agree on every event's listener afterwards."""
    rng = np.random.default_rng(53)
    eng = _engine(max_hosts=2, max_services=256, max_batch_events=1 << 16)
    orc = oracle.OracleEngine(256)
    mid = wire.machine_id(0)
    slot = eng.register_host(mid, "c")
    ns = 77
    A, B = bytes([10, 0, 0, 1]), bytes([10, 0, 0, 2])
    seq = [(1, 80, A), (2, 80, B), (3, 80, None), (4, 80, A),      # 4 replaces 1 (same address), order stays [A', B, any]
           (5, 80, mapped(B)),                                       # ::ffff:B equals B: replaces 2
           (6, 81, None), (7, 81, None),                             # any replaces any
           (8, 82, B), (9, 82, None), (10, 82, None),                # [B, any] then the new any-address one replaces ... B?  no: the first that
           # is any-address OR equal: B is not any and not equal to 0.0.0.0 -> the any-address listener 9 is replaced by 10
           (11, 83, A), (12, 83, B), (13, 83, v6("2001:db8::1")), (14, 83, v6("2001:db8::1"))]
    for g, port, addr in seq:
        a = eng.register_listeners(mid, [g], [ns], [port], addrs=[addr])
        b = orc.register_addr(slot, g, ns, port, addr, is_v6=addr is not None and len(addr) == 16)
        assert a == b
    assert eng.register_listeners(mid, [5], [ns], [80], addrs=[A]) == orc.nsvc  # a known glob_id: nothing new is assigned
    assert eng.num_services() == orc.nsvc == len(seq)
    for rnd in range(3):
        n = 20000
        ev = np.zeros(n, dtype=wire.RESP_EVENT)
        pool = np.array([int.from_bytes(x, "little") for x in (A, B, bytes([10, 0, 0, 3]), bytes(4))], dtype=np.uint32)
        ev["saddr"] = pool[rng.integers(0, 4, n)]
        ev["daddr"] = (0x0B000000 | rng.integers(0, 1 << 12, n)).astype(">u4").view("<u4")
        ev["netns"], ev["sport_be"], ev["dport_be"] = ns, rng.integers(80, 85, n), rng.integers(20000, 60000, n)
        World._times(rng, ev, n)
        eng.handle_resp_events(mid, ev)
        orc.resp_batch(ev.tobytes(), [slot], [0])
        ev6 = np.zeros(n // 2, dtype=wire.RESP_EVENT6)
        sp = np.frombuffer(b"".join([mapped(A), mapped(B), v6("2001:db8::1"), v6("2001:db8::2")]), dtype=np.uint8).reshape(-1, 16)
        ev6["saddr"] = sp[rng.integers(0, 4, n // 2)]
        ev6["daddr"] = np.frombuffer(v6("fd00::9"), dtype=np.uint8)
        ev6["netns"], ev6["sport_be"], ev6["dport_be"] = ns, rng.integers(80, 85, n // 2), rng.integers(20000, 60000, n // 2)
        World._times(rng, ev6, n // 2)
        eng.handle_resp_events_v6(mid, ev6)
        orc.resp_batch_v6(ev6.tobytes(), [slot], [0])
        eng.sync()
        _compare(eng, orc)
    tot = eng.export_hist(0, 0, orc.nsvc)[:, 15, 0]
    replaced = [0, 1, 5, 8, 12]  # slots of glob_ids 1, 2, 6, 9, 13
    assert (tot[replaced] == 0).all() and tot[3] > 0 and tot[4] > 0 and tot[6] > 0 and tot[9] > 0 and tot[13] > 0
    eng.close()


def test_duplicate_listener_records_through_the_device_entry_point(torch_mod, oracle):
    """gys_ingest_listener_state_dev with records that name a listener more than once: every record counts in the host summary
    (LISTEN_SUMM_STATS::update per record, server/gy_mconnhdlr.cc:11252-11258) and the LAST record of a listener in stream order stays as
    its state, whole -- against gyo_listener_state_rollup and a serial walk of the same bytes (not against another ingestion of the library)"""
    torch = torch_mod
    L = oracle.lib()
    rng = np.random.default_rng(54)
    nh, sp = 5, 120
    eng = _engine(max_hosts=8, max_services=nh * sp, enable_tdigest=False)
    info, gids = helpers.register_world(eng, None, range(nh), sp)
    for rnd in range(3):
        recs, hosts = [], []
        for h in range(nh):
            # every listener 0 .. 4 times, in random order; some deletes and bad states among them
            svc = rng.permutation(np.repeat(np.arange(sp), rng.integers(0, 5, sp)))
            r = wire.synth_listener_states(rng, h, svc, delete_frac=0.03, bad_state_frac=0.03)
            r["nqrys_5s"] = rng.integers(0, 1 << 20, len(r))
            recs.append(r)
            hosts.append(np.full(len(r), info[h][1], dtype=np.uint32))
        rec = np.concatenate(recs)
        host = np.concatenate(hosts)
        perm = rng.permutation(len(rec))  # hosts interleaved in one device batch
        rec, host = rec[perm], host[perm]
        tails = [b"x" * int(k) for k in rng.integers(0, 30, len(rec))]  # issue strings: variable stride
        batch = wire.pack_variable(rec, tails)
        raw = np.frombuffer(batch, dtype=np.uint8)
        packed = np.frombuffer(batch, dtype=np.uint8)
        # offsets by walking the packed batch (get_elem_size: 88 + issue_string_len_ + padding_len_)
        offs, o = [], 0
        for i in range(len(rec)):
            offs.append(o)
            o += 88 + int(packed[o + 85]) + int(packed[o + 86])
        assert o == len(packed)
        d = torch.from_numpy(raw.copy()).cuda()
        off = torch.tensor(offs, dtype=torch.int32, device="cuda")
        hostl = torch.from_numpy(host.view(np.int32).copy()).cuda()
        eng.order()
        capi.check(eng.L.gys_ingest_listener_state_dev(eng.h, C.c_void_p(d.data_ptr()), C.c_void_p(off.data_ptr()), C.c_void_p(hostl.data_ptr()), len(rec)))
        eng.sync()
        # ---- the oracle's serial walk of the same bytes
        summ = {h: oracle.ListenSummStats() for h in range(nh)}
        kept = {}
        slot_of_host = {info[h][1]: h for h in range(nh)}
        for i in range(len(rec)):
            h = slot_of_host[int(host[i])]
            r88 = packed[offs[i]:offs[i] + 88]
            g = int(rec["glob_id"][i])
            if rec["query_flags"][i] == wire.LISTEN_FLAG_DELETE:
                kept.pop(g, None)
                continue
            if rec["curr_state"][i] > 5:
                continue
            nerr = C.c_int(0)
            buf = np.ascontiguousarray(r88)
            L.gyo_listener_state_rollup(buf.ctypes.data, 1, buf.ctypes.data + 88, C.byref(summ[h]), C.byref(nerr))
            kept[g] = r88.tobytes()
        # kept state of every listener, as the filtered query returns it (no criteria: every current record)
        gs, gh, gr, nm = eng.svcstate_scan(maxrecs=nh * sp)
        got = {int(gr[k]["glob_id"]): gr[k].tobytes() for k in range(len(gs))}
        assert set(got) == set(kept), (len(got), len(kept))
        for g, b in kept.items():
            # (bytes 85 / 86: issue string and padding lengths of the message, not part of the state)
            gb, wb = bytearray(got[g]), bytearray(b)
            gb[85:88] = wb[85:88] = b"\0\0\0"
            assert gb == wb, "kept state of listener %x differs from the last record of the stream" % g
        eng.window_close()
        for h in range(nh):
            assert eng.svcsumm(info[h][0]).as_tuple() == summ[h].as_tuple(), "host %d summary" % h
        eng.window_close()  # (a state stays current for the window after its own: the next round starts two windows on)
    eng.close()


@pytest.mark.parametrize("cap", [1920, 3968, 3072])
def test_larger_digest_buffers_equal_the_oracle_with_the_same_buffer(torch_mod, oracle, cap):
    """gys_config.td_pend_cap: the digests buffer more values between re-clusterings (merges of up to 2048 / 4096 values through the larger
    instances of the value-bin kernel).  State after every round == the oracle engine built with the same buffer size; the all-service
    quantile scan, a roll-up digest and single-service quantiles read the larger buffers as well."""
    torch = torch_mod
    rng = np.random.default_rng(60 + cap)
    nh, sp = 3, 40
    eng = _engine(max_hosts=4, max_services=nh * sp, max_batch_events=1 << 20, td_pend_cap=cap)
    orc = oracle.OracleEngine(nh * sp, td_cap=cap)
    assert eng.L.gys_td_pend_cap(eng.h) == cap
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    L = oracle.lib()
    merges0 = 0
    for rnd in range(9):
        for h in range(nh):
            # ~400 .. 1600 values per service and call: several calls fill a buffer, some calls overflow the fast merge class at once (class 1 / several-workgroup paths)
            n = int(rng.integers(400, 1600)) * sp if rnd != 6 else 5000 * sp
            ev = helpers.make_resp_events(rng, h, n, sp, lat_mu=3.0 + 0.8 * h, lat_sigma=1.7)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
        eng.sync()
        n = orc.nsvc
        gs, gc, gm = eng.export_tdigest(0, n)
        os_, oc, om = orc.td_arrays()
        assert (gc == oc).all() and (gs == os_).all() and (gm == om).all(), "round %d" % rnd
        gn, gp = eng.export_tdigest_pending(0, n)
        on, op = orc.td_pending()
        assert gp.shape == op.shape == (n, cap)
        assert (gn == on).all() and (gp == op).all()
        helpers.assert_hist_equal(eng.export_hist(0, 0, n), orc.hist(), n)
    assert int(gn.max()) > 1024 or cap < 1024  # the larger buffers are really in use
    c = eng.counters()
    assert c["td_merges"] > 0
    # single-service quantiles (merged view through the query kernel) and the all-service scan (k_digest_bins<SCAN, VPT>)
    qs = [0.25, 0.5, 0.95, 0.99]
    for h in range(nh):
        for s in (0, sp // 2, sp - 1):
            g = int(gids[h][s])
            assert eng.quantiles(g, qs) == [L.gyo_tdb_quantile(C.byref(orc.td(eng.lookup(g))), q) for q in qs]
    got = np.asarray(eng.scan_quantiles(qs)).reshape(orc.nsvc, len(qs))
    want = np.array([[L.gyo_tdb_quantile(C.byref(orc.td(s)), q) for q in qs] for s in range(orc.nsvc)])
    assert (got == want).all()
    eng.close()


def test_level0_is_the_swapped_window_array(torch_mod, oracle):
    """enable_levels = 1 with lazily folded records: the close leaves level 0 behind by SWAPPING the window-record array with the level-0 array
    (per-service window tags written by the close's fold pass, level_roll in gys_engine.hip) instead of copying a record per service.  What has
    to hold whatever a service did in which window: level 0 = the closed window's record (empty for a service silent in it) right after the
    close, in the prepared state, and still after the NEXT window's events have been folded into the other array by queries and merges; the
    window view = the open window's record so far; the all-time view = everything.  Services alternate between silent and busy windows so that
    both arrays hold stale records of different ages; enough values per call for merges (folds inside the merge kernels) in between."""
    rng = np.random.default_rng(905)
    nh, sp = 2, 7
    nsvc = nh * sp
    eng = _engine(max_hosts=4, max_services=32, max_batch_events=1 << 16, enable_levels=True)
    orc_win = oracle.OracleEngine(32, enable_td=False)
    orc_all = oracle.OracleEngine(32, enable_td=False)
    info, gids = helpers.register_world(eng, orc_win, range(nh), sp)
    helpers.register_world(None, orc_all, range(nh), sp)
    t = 1_700_000_000
    empty = np.zeros((nsvc, 15, 2), dtype=np.int64)
    prev = empty

    def feed(w, part):
        for h in range(nh):
            active = [s for s in range(sp) if (w + s + h) % 3 != 0 and not (w % 4 == 3 and h == 1)]  # silent services, a silent host every 4th window
            if not active:
                continue
            n = int(rng.integers(200, 900)) * len(active) if (w + part) % 3 else 2600 * len(active)  # (2600 per service: past the buffer, merges)
            ev = helpers.make_resp_events(rng, h, n, sp, lat_mu=2.5 + 0.2 * (w % 5))
            s_idx = np.array(active)[rng.integers(0, len(active), n)]
            ev["netns"] = wire.listener_netns(h, s_idx)
            ev["sport_be"] = wire.listener_port(s_idx)
            if (w + part + h) % 4 == 3:
                # the same traffic as IPv6 events (handle_ipv6_resp_event adds to the same resp_hist_, gy_socket_stat.cc:1577-1592)
                e6 = np.zeros(n, dtype=wire.RESP_EVENT6)
                e6["saddr"][:, 0], e6["saddr"][:, 1], e6["saddr"][:, 15] = 0x20, 0x01, h + 1
                e6["daddr"][:, 0] = 0xFD
                e6["daddr"][:, 12:16] = rng.integers(0, 256, (n, 4))
                for f in ("netns", "sport_be", "dport_be", "lsndtime", "lrcvtime"):
                    e6[f] = ev[f]
                eng.handle_resp_events_v6(info[h][0], e6)
                for o in (orc_win, orc_all):
                    o.resp_batch_v6(e6.tobytes(), [info[h][1]], [0])
                continue
            eng.handle_resp_events(info[h][0], ev)
            for o in (orc_win, orc_all):
                o.resp_batch(ev.tobytes(), [info[h][1]], [0])

    def lv0(tq):
        return eng.export_hist_level(0, tq * 1_000_000, 0, nsvc)[:, :15, :]

    for w in range(14):
        t += 5
        feed(w, 0)
        # mid-window: the queries fold the open window into the (swapped-in) window array; level 0 still is the window closed before
        assert (lv0(t - 3) == prev).all(), "window %d: level 0 after the next window's first folds" % w
        helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc_win.hist(), nsvc)
        helpers.assert_hist_equal(eng.export_hist(1, 0, nsvc), orc_all.hist(), nsvc)
        feed(w, 1)
        assert (lv0(t - 1) == prev).all(), "window %d: level 0 before the close" % w
        win = np.array(orc_win.hist()[:nsvc])[:, :15, :]
        if w % 3 == 1:
            capi.check(eng.L.gys_window_prepare(eng.h, t * 1_000_000))
            # prepared, not finished: the closing window is level 0 already and still is what the window view shows
            assert (lv0(t) == win).all(), "window %d: level 0 in the prepared state" % w
            helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc_win.hist(), nsvc)
            capi.check(eng.L.gys_window_finish(eng.h))
        else:
            eng.window_close(t * 1_000_000)
        assert (lv0(t) == win).all(), "window %d: level 0 after the close" % w
        assert (lv0(t + 4) == win).all()
        assert (lv0(t + 5) == empty).all()  # a 5-s ring keeps an add for 5 s
        orc_win.window_clear(clear_hist=True)
        orc_all.window_clear(clear_hist=False)
        helpers.assert_hist_equal(eng.export_hist(0, 0, nsvc), orc_win.hist(), nsvc)  # the new window is empty for every service
        helpers.assert_hist_equal(eng.export_hist(1, 0, nsvc), orc_all.hist(), nsvc)
        prev = win
    assert eng.counters()["td_merges"] > 0
    # a service registered after many closes: no level-0 record until a window of its own has closed (its tag names no window yet)
    gid_new, ns_new, port_new = 0x7777, int(wire.listener_netns(0, 0)), 20000
    s_new = eng.register_listeners(info[0][0], [gid_new], [ns_new], [port_new])
    for o in (orc_win, orc_all):
        assert o.register(info[0][1], gid_new, ns_new, port_new) == s_new
    assert s_new == nsvc
    assert (eng.export_hist_level(0, t * 1_000_000, 0, nsvc + 1)[nsvc, :15, :] == 0).all()
    for rnd in range(2):
        t += 5
        ev = helpers.make_resp_events(rng, 0, 700, sp, unknown_frac=0.0)
        ev["netns"], ev["sport_be"] = ns_new, port_new
        eng.handle_resp_events(info[0][0], ev)
        for o in (orc_win, orc_all):
            o.resp_batch(ev.tobytes(), [info[0][1]], [0])
        eng.window_close(t * 1_000_000)
        win = np.array(orc_win.hist()[:nsvc + 1])[:, :15, :]
        assert win[nsvc, :, 0].sum() > 600
        assert (eng.export_hist_level(0, t * 1_000_000, 0, nsvc + 1)[:, :15, :] == win).all(), "late service, close %d" % rnd
        orc_win.window_clear(clear_hist=True)
        orc_all.window_clear(clear_hist=False)
    helpers.assert_hist_equal(eng.export_hist(1, 0, nsvc + 1), orc_all.hist(), nsvc + 1)
    eng.close()
