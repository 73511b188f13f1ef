"""wire-record layouts (gyeeta_amd/wire.py and the oracle's byte-offset decoder) against what the reference's own compiled
IP_PORT produces (golden fixture) and against each other."""
import json
import os

import numpy as np

from gyeeta_amd import wire

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.json")


def test_ip_port_layout_matches_reference_object_bytes():
    g = json.load(open(GOLD))
    assert g["sizes"] == {"GY_IP_ADDR": 24, "IP_PORT": 32, "PAIR_IP_PORT": 64, "NS_IP_PORT": 40, "HIST_SERIAL": 16,
                          "GY_HISTOGRAM_RESP": 280, "HIST_DATA": 32}
    a = np.zeros(1, dtype=wire.IP_PORT)
    wire.set_ip_port(a, ip32_be=np.uint32(0x0100007F), port=8080)
    ref = bytes.fromhex(g["ip_port_layout_v4"])
    got = a.tobytes()
    assert got[:22] == ref[:22] and got[24:26] == ref[24:26]  # ipflags_ (bytes 22-23) is derived metadata, never hashed
    b = np.zeros(1, dtype=wire.IP_PORT)
    wire.set_ip_port(b, ip128=np.arange(1, 17, dtype=np.uint8), port=443)
    ref6 = bytes.fromhex(g["ip_port_layout_v6"])
    assert b.tobytes()[:22] == ref6[:22] and b.tobytes()[24:26] == ref6[24:26]


def test_variable_stride_walk(oracle):
    rng = np.random.default_rng(2)
    L = oracle.lib()
    rec = wire.synth_tcp_conns(rng, 50, [0, 1], 5, v6_frac=0.3)
    tails = [bytes([65] * int(k)) for k in rng.integers(0, 257, 50)]
    batch = wire.pack_variable(rec, tails)
    buf = np.frombuffer(batch, dtype=np.uint8)
    off = 0
    for i in range(50):
        sz = L.gyo_tcp_conn_elem_size(buf.ctypes.data + off)
        assert sz % 8 == 0 and sz == 280 + len(tails[i]) + (-(280 + len(tails[i])) % 8)
        assert bytes(buf[off + 280:off + 280 + len(tails[i])]) == tails[i]
        off += sz
    assert off == len(batch)
    kw = np.zeros(500, dtype=np.uint32)
    nw = np.zeros(50, dtype=np.uint32)
    gid = np.zeros(50, dtype=np.uint64)
    bs = np.zeros(50, dtype=np.uint64)
    br = np.zeros(50, dtype=np.uint64)
    n = L.gyo_tcp_conn_decode(buf.ctypes.data, 50, buf.ctypes.data + len(buf), oracle.ptr(kw, oracle.u32p), oracle.ptr(nw, oracle.u32p),
                              oracle.ptr(gid, oracle.u64p), oracle.ptr(bs, oracle.u64p), oracle.ptr(br, oracle.u64p), None)
    assert n == 50 and (gid == rec["ser_glob_id"]).all() and (bs == rec["bytes_sent"]).all()
    for i in range(50):
        c6 = int(rec["nat_cli"]["ip32_be"][i]) == 0
        cip, cf = oracle.ip_bytes(bytes(rec["nat_cli"]["ip128"][i]) if c6 else int(rec["nat_cli"]["ip32_be"][i]))
        sip, sf = oracle.ip_bytes(int(rec["nat_ser"]["ip32_be"][i]))
        exp = L.gyo_pair_ip_port_hash(cip, cf, int(rec["nat_cli"]["port"][i]), sip, sf, int(rec["nat_ser"]["port"][i]))
        w = kw[i * 10:i * 10 + nw[i]]
        assert L.gyo_jhash2(oracle.ptr(np.ascontiguousarray(w), oracle.u32p), int(nw[i]), 0xCEEDFEAD) == exp
        assert nw[i] == (7 if c6 else 4)
    # pend earlier than the batch end stops the walk like the reference loop condition (uint8_t*)p < pendptr
    n2 = L.gyo_tcp_conn_decode(buf.ctypes.data, 50, buf.ctypes.data + 280, oracle.ptr(kw, oracle.u32p), oracle.ptr(nw, oracle.u32p),
                               oracle.ptr(gid, oracle.u64p), oracle.ptr(bs, oracle.u64p), oracle.ptr(br, oracle.u64p), None)
    assert n2 == 1


# ---------------------------------------------------------------------------------------------------------------------------------
# the reference's OWN wire structs and validators (common/gy_comm_proto.h / .cc compiled into oracle/_ref by oracle/build_ref.sh)
import ctypes as C  # noqa: E402

import pytest  # noqa: E402


def _aligned(b, extra=0):
    """8-byte aligned, writable copy of a message (the reference's record validators NUL-terminate strings in place)"""
    a = np.zeros((len(b) + extra + 7) // 8 + 1, dtype=np.uint64)
    v = a.view(np.uint8)
    v[:len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a, v


def test_wire_layouts_vs_reference_structs(reflib):
    """every member offset of the numpy layouts in gyeeta_amd/wire.py (and of gys_listener_day_stats) == offsetof in the reference's
    structs as the compiler lays them out; sizes; the framing constants the engine and wire.py use"""
    R = reflib
    if not hasattr(R, "ref_comm_sizeof"):
        pytest.skip("oracle/_ref built without gy_comm_proto")
    assert [R.ref_comm_sizeof(i) for i in range(6)] == [16, 8, wire.TCP_CONN_NOTIFY.itemsize, wire.LISTENER_STATE_NOTIFY.itemsize, 48, 16]
    from gyeeta_amd import capi
    day = {n: getattr(capi.ListenerDayStats, n).offset for n, _ in capi.ListenerDayStats._fields_}
    hdr = {"magic": 0, "total_sz": 4, "data_type": 8, "padding_sz": 12}   # wire.frame_event_notify / gys_ingest_comm_stream
    evn = {"subtype": 0, "nevents": 4}
    seen = set()
    for i in range(R.ref_comm_nfields()):
        name, off = R.ref_comm_field_name(i).decode(), R.ref_comm_field_offset(i)
        st, f = name.split(".")
        if st == "TCP_CONN_NOTIFY":
            assert wire.TCP_CONN_NOTIFY.fields[f][1] == off, name
        elif st == "LISTENER_STATE_NOTIFY":
            assert wire.LISTENER_STATE_NOTIFY.fields[f][1] == off, name
        elif st == "LISTENER_DAY_STATS":
            assert day[f] == off, name
        elif st == "ACTIVE_CONN_STATS":
            assert wire.ACTIVE_CONN_STATS.fields[f][1] == off, name
        elif st == "COMM_HEADER":
            assert hdr[f] == off, name
        else:
            assert st == "EVENT_NOTIFY" and evn[f] == off, name
        seen.add(name)
    # nothing of the numpy layouts is left unchecked (tail_pad is the struct's alignment padding)
    assert {"TCP_CONN_NOTIFY." + n for n in wire.TCP_CONN_NOTIFY.names} <= seen
    assert {"LISTENER_STATE_NOTIFY." + n for n in wire.LISTENER_STATE_NOTIFY.names if n != "tail_pad"} <= seen
    assert {"LISTENER_DAY_STATS." + n for n in day} <= seen
    assert {"ACTIVE_CONN_STATS." + n for n in wire.ACTIVE_CONN_STATS.names if n not in ("flags", "tail_pad")} <= seen
    # the three 1-bit flags behind active_conns_: byte offset and bit values as the reference's compiler lays the bit-fields out
    if hasattr(R, "ref_active_conn_flag"):
        import ctypes as C
        R.ref_active_conn_flag.restype, R.ref_active_conn_flag.argtypes = C.c_uint32, [C.c_int]
        off = wire.ACTIVE_CONN_STATS.fields["flags"][1]
        assert [R.ref_active_conn_flag(i) for i in range(3)] == [(off << 8) | wire.ACTIVE_FLAG_CLI_LISTENER_PROC, (off << 8) | wire.ACTIVE_FLAG_REMOTE_LISTEN,
                                                                 (off << 8) | wire.ACTIVE_FLAG_REMOTE_CLI]
        assert R.ref_comm_sizeof(6) == wire.ACTIVE_CONN_STATS.itemsize == 104
    want = [wire.PM_HDR_MAGIC, wire.COMM_EVENT_NOTIFY, 1, 18, 16 << 20, wire.NOTIFY_TCP_CONN, wire.NOTIFY_LISTENER_STATE, 2048, 512, 2048,
            wire.LISTEN_FLAG_DELETE]
    assert [R.ref_comm_const(i) for i in range(11)] == want


def _mutations(rng, good, rec_size):
    """malformed variants of a well-formed EVENT_NOTIFY message: header fields, nevents_, length fields of records"""
    out = []
    for _ in range(60):
        b = bytearray(good)
        kind = int(rng.integers(0, 9))
        if kind == 0:
            b[0:4] = int(rng.choice([0x05777705, 0, 0x05666606])).to_bytes(4, "little")          # other / no magic
        elif kind == 1:
            b[4:8] = int(rng.choice([len(good) + 4, len(good) - 4, 8, 12, (16 << 20), (16 << 20) + 8, 0])).to_bytes(4, "little")
        elif kind == 2:
            b[8:12] = int(rng.choice([0, 1, 18, 19, 255, 14, 15])).to_bytes(4, "little")           # data_type_ range
        elif kind == 3:
            b[12:16] = int(rng.choice([0, 7, 8, 9, 200])).to_bytes(4, "little")                    # padding_sz_
        elif kind == 4:
            b[20:24] = int(rng.choice([0, 1, 2047, 2048, 2049, 511, 512, 513, 2**31])).to_bytes(4, "little")  # nevents_
        elif kind == 5:                                                                            # a record's variable-length fields
            off = 24
            if rec_size == 280:
                b[off + 272:off + 274] = int(rng.integers(0, 5000)).to_bytes(2, "little")
                b[off + 279] = int(rng.integers(0, 9))
            else:
                b[off + 85] = int(rng.integers(0, 256))
                b[off + 86] = int(rng.integers(0, 9))
        elif kind == 6:                                                                            # truncated message
            cut = int(rng.integers(3, len(good) // 8)) * 8
            b = b[:cut]
            b[4:8] = cut.to_bytes(4, "little")
        elif kind == 7:                                                                            # random byte flips in the headers
            for k in rng.integers(0, 24, 2):
                b[int(k)] ^= int(rng.integers(1, 256))
        out.append(bytes(b))                                                                       # kind 8: unchanged
    return out


@pytest.mark.parametrize("which", ["tcp_conn", "listener_state"])
def test_l1_validators_vs_reference(reflib, oracle, which):
    """COMM_HEADER::validate + TCP_CONN_NOTIFY::validate / LISTENER_STATE_NOTIFY::validate: the oracle's restatement (the CPU twin of the
    engine's wire front-end) against the reference's own functions, on well-formed messages built by wire.py and on malformed variants"""
    R, L = reflib, oracle.lib()
    if not hasattr(R, "ref_comm_hdr_validate"):
        pytest.skip("oracle/_ref built without gy_comm_proto")
    rng = np.random.default_rng(41 if which == "tcp_conn" else 43)
    accepted = rejected = 0
    for trial in range(25):
        n = int(rng.integers(1, 60))
        if which == "tcp_conn":
            rec = wire.synth_tcp_conns(rng, n, [0, 1], 7, v6_frac=0.2)
            tails = [bytes(rng.integers(1, 255, int(k), dtype=np.uint8).tolist()) for k in rng.integers(0, 200, n)]
            sub, rsz = wire.NOTIFY_TCP_CONN, 280
            rv, ov = R.ref_tcp_conn_validate, L.gyo_tcp_conn_validate
        else:
            rec = wire.synth_listener_states(rng, 0, rng.integers(0, 50, n))
            tails = [b"i" * int(k) for k in rng.integers(0, 120, n)]
            sub, rsz = wire.NOTIFY_LISTENER_STATE, 88
            rv, ov = R.ref_listener_state_validate, L.gyo_listener_state_validate
        good = wire.frame_event_notify(sub, n, wire.pack_variable(rec, tails))
        for msg in [good] + _mutations(rng, good, rsz):
            keep, v = _aligned(msg, extra=8192)   # slack: a lying length field must not send either validator out of the buffer
            p = v.ctypes.data
            h_ref, h_orc = R.ref_comm_hdr_validate(p, wire.PM_HDR_MAGIC), L.gyo_comm_header_validate(p, wire.PM_HDR_MAGIC)
            dt = int.from_bytes(msg[8:12], "little")
            if dt == wire.COMM_EVENT_NOTIFY or not h_ref:
                assert h_ref == h_orc, (trial, msg[:24].hex())
            if h_ref and dt == wire.COMM_EVENT_NOTIFY and int.from_bytes(msg[4:8], "little") <= len(msg):
                before = bytes(v[:len(msg)])
                r = rv(p)
                v[:len(msg)] = np.frombuffer(before, dtype=np.uint8)  # undo the reference's in-place NUL termination
                o = ov(p)
                assert r == o, (trial, msg[:24].hex())
                accepted += r
                rejected += 1 - r
        assert R.ref_comm_hdr_validate(_aligned(good)[1].ctypes.data, wire.PM_HDR_MAGIC) == 1
        assert R.ref_comm_hdr_validate(_aligned(good)[1].ctypes.data + 4, wire.PM_HDR_MAGIC) == 0  # unaligned data is refused
        # record sizes: wire.pack_variable pads like set_padding_len, get_elem_size agrees
        buf = np.frombuffer(good, dtype=np.uint8)
        off = 24
        for i in range(n):
            rs = (R.ref_tcp_conn_elem_size if rsz == 280 else R.ref_listener_state_elem_size)(buf.ctypes.data + off)
            os_ = (L.gyo_tcp_conn_elem_size if rsz == 280 else L.gyo_listener_state_elem_size)(buf.ctypes.data + off)
            assert rs == os_ and rs % 8 == 0
            off += rs
        assert off == len(good) - int.from_bytes(good[12:16], "little")  # the records end where the message's padding_sz_ bytes begin
    assert accepted > 50 and rejected > 50


def test_malformed_streams_of_the_gpu_test_are_malformed_for_the_reference(reflib, oracle):
    """the hand-made malformed messages tests/test_gpu_wire.py feeds to gys_ingest_comm_stream (expecting a rejection) are rejected by the
    reference's validators as well, and the well-formed one is accepted"""
    R, L = reflib, oracle.lib()
    if not hasattr(R, "ref_comm_hdr_validate"):
        pytest.skip("oracle/_ref built without gy_comm_proto")
    rng = np.random.default_rng(9)
    n = 40
    rec = wire.synth_tcp_conns(rng, n, [0], 12)
    tails = [bytes([97 + int(k) % 26] * int(k)) for k in rng.integers(0, 64, n)]
    payload = wire.pack_variable(rec, tails)
    good = wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload)

    def verdict(msg):
        keep, v = _aligned(msg, extra=8192)
        p = v.ctypes.data
        ref = bool(R.ref_comm_hdr_validate(p, wire.PM_HDR_MAGIC)) and int.from_bytes(msg[4:8], "little") <= len(msg) and bool(R.ref_tcp_conn_validate(p))
        keep2, v2 = _aligned(msg, extra=8192)
        p2 = v2.ctypes.data
        orc = bool(L.gyo_comm_header_validate(p2, wire.PM_HDR_MAGIC)) and int.from_bytes(msg[4:8], "little") <= len(msg) and bool(L.gyo_tcp_conn_validate(p2))
        assert ref == orc
        return ref

    assert verdict(good)
    assert not verdict(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload, magic=0x05777705))
    assert not verdict(good[:4] + np.array([len(good) + 4], dtype="<u4").tobytes() + good[8:])
    assert not verdict(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, 2049, payload))
    assert not verdict(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n + 1, payload))
    bad = bytearray(payload)
    bad[272:274] = (3).to_bytes(2, "little")
    bad[279] = 0
    assert not verdict(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, bytes(bad)))
    bad = bytearray(payload)
    bad[272:274] = (4000).to_bytes(2, "little")
    assert not verdict(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, bytes(bad)))
