"""Row a7 of SURVEY 8: the listener lookup of a response event compares the server ADDRESS (operator==(shared_ptr<TCP_LISTENER>, NS_IP_PORT),
common/gy_socket_stat.h:708-714) and IPv6 events build their addresses through GY_IP_ADDR(unsigned __int128) (common/gy_socket_stat.cc:1535-1551).
The oracle's restatement (gyo_ip_norm / gyo_ip_equal / the engine's chains) is pinned here to the reference's own GY_IP_ADDR / NS_IP_PORT /
PAIR_IP_PORT compiled in place (oracle/_ref), incl. the addresses that EMBED an IPv4 one: ip32_be_ and embedded_ipv4_ share their storage
(common/gy_common_inc.h:10497-10500), so ::ffff:a.b.c.d, 2002::/16 and 64:ff9b::/32 addresses hash and compare as IPv4."""
import ctypes as C
import ipaddress

import numpy as np
import pytest

SPECIAL_V6 = ["::", "::1", "::2", "::ffff:10.1.2.3", "::ffff:0.0.0.0", "::fffe:10.1.2.3", "2002:0a01:0203::1", "2002::", "2001:db8::5", "2001:470::9",
              "2a02:26f0::1", "64:ff9b::10.1.2.3", "64:ff9b:1::10.1.2.3", "64:ff9a::10.1.2.3", "fe80::1", "fd00::7", "fc00::8", "ff02::1", "::10.1.2.3",
              "0:0:0:1::ffff:0a01:0203", "1::ffff:10.1.2.3", "2002:0a01:0203:ffff:ffff:ffff:ffff:ffff", "3000::1", "2fff::1"]
SPECIAL_V4 = ["0.0.0.0", "10.1.2.3", "127.0.0.1", "255.255.255.255", "0.0.0.1", "1.0.0.0"]


def addr_cases(rng, nrand=300):
    cases = [(ipaddress.IPv6Address(a).packed, 1) for a in SPECIAL_V6] + [(ipaddress.IPv4Address(a).packed, 0) for a in SPECIAL_V4]
    for _ in range(nrand):
        cases.append((rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), 1))
        cases.append((rng.integers(0, 256, 4, dtype=np.uint8).tobytes(), 0))
        b = bytearray(16)  # sparse random bytes: hits the prefix checks from both sides
        for _k in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, 16))] = int(rng.choice([0x00, 0x01, 0x02, 0x20, 0x64, 0x9B, 0xFF, 0xFE, 0x0A]))
        cases.append((bytes(b), 1))
    return cases


def u8(b):
    return (C.c_uint8 * 16)(*(bytes(b) + bytes(16))[:16])


def norm_oracle(L, ip, v6):
    ip32, ip128 = C.c_uint32(), (C.c_uint8 * 16)()
    is_any = L.gyo_ip_norm(u8(ip), v6, C.byref(ip32), ip128)
    return ip32.value, bytes(ip128), is_any


def test_ip_norm_equals_reference(oracle, reflib):
    L, R = oracle.lib(), reflib
    if not hasattr(R, "ref_ip_addr_norm"):
        pytest.skip("oracle/_ref predates ref_ip_addr_norm")
    rng = np.random.default_rng(5)
    cases = addr_cases(rng)
    for ip, v6 in cases:
        o32, o128, oany = norm_oracle(L, ip, v6)
        r32, r128, rin, rlen = C.c_uint32(), (C.c_uint8 * 16)(), (C.c_uint8 * 16)(), C.c_uint32()
        rany = R.ref_ip_addr_norm(u8(ip), v6, C.byref(r32), r128, rin, C.byref(rlen))
        assert (o32, o128, oany) == (r32.value, bytes(r128), rany), (ip.hex(), v6)
        # what the hashes see: 4 bytes when ip32_be_ != 0 else the 16 bytes (get_as_inaddr)
        want = (r32.value.to_bytes(4, "little"), 4) if r32.value else (bytes(r128), 16)
        assert (bytes(rin)[:rlen.value], rlen.value) == want
    # the embedded forms really are recognised (guards against a vacuous test)
    assert norm_oracle(L, ipaddress.IPv6Address("::ffff:10.1.2.3").packed, 1)[0] == int.from_bytes(bytes([10, 1, 2, 3]), "little")
    assert norm_oracle(L, ipaddress.IPv6Address("2002:0a01:0203::1").packed, 1)[0] == int.from_bytes(bytes([10, 1, 2, 3]), "little")
    assert norm_oracle(L, ipaddress.IPv6Address("64:ff9b::10.1.2.3").packed, 1)[0] == int.from_bytes(bytes([10, 1, 2, 3]), "little")
    assert norm_oracle(L, ipaddress.IPv6Address("2001:db8::5").packed, 1)[0] == 0


def test_equality_match_and_flow_hash_equal_reference(oracle, reflib):
    L, R = oracle.lib(), reflib
    if not hasattr(R, "ref_listener_match"):
        pytest.skip("oracle/_ref predates ref_listener_match")
    rng = np.random.default_rng(6)
    cases = addr_cases(rng, nrand=60)
    # pairs that are equal only through the embedded address
    m = ipaddress.IPv6Address("::ffff:10.1.2.3").packed
    cases += [(m, 1), (ipaddress.IPv4Address("10.1.2.3").packed, 0), (ipaddress.IPv6Address("2002:0a01:0203::77").packed, 1)]
    idx = rng.integers(0, len(cases), (4000, 2))
    neq = 0
    for i, j in idx:
        (a, a6), (b, b6) = cases[i], cases[j]
        a32, a128, _ = norm_oracle(L, a, a6)
        b32, b128, _ = norm_oracle(L, b, b6)
        oeq = L.gyo_ip_equal(a32, u8(a128), b32, u8(b128))
        assert oeq == R.ref_ip_addr_equal(u8(a), a6, u8(b), b6), (a.hex(), b.hex())
        neq += oeq
        # operator==(listener, NS_IP_PORT): listener (a) any / bound, event (b); inode and port equal or not
        for l_any in (0, 1):
            for dport, dns in ((0, 0), (1, 0), (0, 1)):
                want = R.ref_listener_match(u8(a), a6, 8080, 4026531840, l_any, u8(b), b6, 8080 + dport, 4026531840 + dns)
                got = int(dport == 0 and dns == 0 and (l_any or oeq))
                assert got == want
        # flow key of a response event: PAIR_IP_PORT(cli = daddr:dport, ser = saddr:sport)
        w = np.zeros(10, dtype=np.uint32)
        nw = L.gyo_pair_ip_port_words(u8(a), a6, 40000, u8(b), b6, 443, oracle.ptr(w, oracle.u32p))
        assert L.gyo_jhash2(oracle.ptr(w, oracle.u32p), nw, 0xceedfead) == R.ref_pair_ip_port_hash(u8(a), a6, 40000, u8(b), b6, 443)
    assert neq > 20


def test_engine_lookup_is_first_match_in_registration_order(oracle):
    """the oracle engine's chains against a direct statement of the rule on a small world (listener_tbl_ lookup, common/gy_socket_stat.cc:1671)"""
    from gyeeta_amd import wire
    rng = np.random.default_rng(9)
    orc = oracle.OracleEngine(64, enable_td=False)
    A, B, M = bytes([10, 0, 0, 1]), bytes([10, 0, 0, 2]), ipaddress.IPv6Address("::ffff:10.0.0.1").packed
    V6 = ipaddress.IPv6Address("2001:db8::1").packed
    # key (ns 7, port 80): bound A, then bound V6, then any; key (7, 81): any then bound B (replaces the any one in place); key (7, 82): bound B only
    s_a = orc.register_addr(0, 100, 7, 80, A)
    s_v6 = orc.register_addr(0, 101, 7, 80, V6, is_v6=True)
    s_any = orc.register_addr(0, 102, 7, 80)
    s_old = orc.register_addr(0, 103, 7, 81)
    s_new = orc.register_addr(0, 104, 7, 81, B)
    s_b = orc.register_addr(0, 105, 7, 82, B)
    s_m = orc.register_addr(0, 106, 7, 83, M, is_v6=True)  # bound to ::ffff:10.0.0.1 == 10.0.0.1

    def one_v4(saddr, port):
        ev = np.zeros(1, dtype=wire.RESP_EVENT)
        ev["saddr"] = int.from_bytes(saddr, "little")
        ev["daddr"] = int.from_bytes(bytes([10, 9, 9, 9]), "little")
        ev["netns"] = 7
        ev["sport_be"] = port
        ev["dport_be"] = 40000
        ev["lrcvtime"] = 1000
        ev["lsndtime"] = 1010
        return ev

    def hit(ev, v6=False):
        before = orc.hist()[:, 15, 0].copy()
        (orc.resp_batch_v6 if v6 else orc.resp_batch)(ev.tobytes(), [0], [0])
        d = orc.hist()[:, 15, 0] - before
        assert d.sum() <= 1
        return int(np.argmax(d)) if d.sum() else None

    assert hit(one_v4(A, 80)) == s_a
    assert hit(one_v4(B, 80)) == s_any            # nobody bound to B on 80: the any-address listener behind the bound ones
    assert hit(one_v4(A, 81)) is None and hit(one_v4(B, 81)) == s_new  # the bound listener REPLACED the any-address one
    assert s_old != s_new
    assert hit(one_v4(A, 82)) is None and hit(one_v4(B, 82)) == s_b
    assert hit(one_v4(A, 83)) == s_m              # the IPv6 listener address embeds 10.0.0.1
    ev6 = np.zeros(1, dtype=wire.RESP_EVENT6)
    ev6["saddr"] = np.frombuffer(V6, dtype=np.uint8)
    ev6["daddr"] = np.frombuffer(ipaddress.IPv6Address("2001:db8::99").packed, dtype=np.uint8)
    ev6["netns"], ev6["sport_be"], ev6["dport_be"], ev6["lrcvtime"], ev6["lsndtime"] = 7, 80, 40001, 5, 25
    assert hit(ev6, v6=True) == s_v6
    ev6["saddr"] = np.frombuffer(M, dtype=np.uint8)
    assert hit(ev6, v6=True) == s_a               # ::ffff:10.0.0.1 on port 80 -> the listener bound to 10.0.0.1
    bm = orc.bitmap()
    assert bm[s_a, :32].any() and bm[s_a, 32:].any() and not bm[s_v6, :32].any() and bm[s_v6, 32:].any()
