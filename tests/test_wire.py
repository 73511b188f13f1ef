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
