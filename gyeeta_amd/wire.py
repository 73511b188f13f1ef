"""numpy layouts of the reference's little-endian, 8-byte aligned wire records on this path (common/gy_comm_proto.h) and
deterministic synthetic stream builders.  Host-side plumbing for tests / bench; the engine itself consumes the
raw bytes."""
import numpy as np

AF_INET, AF_INET6 = 2, 10

# class IP_PORT (common/gy_common_inc.h:11162+): GY_IP_ADDR {ip128_be_ @0, ip32_be_ @16, aftype_ @20, ipflags_ @22} (24 B) + port_ @24, 32 B
IP_PORT = np.dtype([("ip128", "u1", 16), ("ip32_be", "<u4"), ("aftype", "<i2"), ("ipflags", "<u2"), ("port", "<u2"), ("pad", "u1", 6)])
assert IP_PORT.itemsize == 32

# struct TCP_CONN_NOTIFY (common/gy_comm_proto.h:1665-1742): 280 fixed bytes (+ cli_cmdline_len_ + padding_len_)
TCP_CONN_NOTIFY = np.dtype([
    ("cli", IP_PORT), ("ser", IP_PORT), ("nat_cli", IP_PORT), ("nat_ser", IP_PORT),
    ("tusec_start", "<u8"), ("tusec_close", "<u8"), ("cli_task_aggr_id", "<u8"), ("cli_related_listen_id", "<u8"),
    ("cli_madhava_id", "<u8"), ("cli_ser_machine_id", "<u8", 2), ("ser_related_listen_id", "<u8"), ("ser_glob_id", "<u8"),
    ("ser_madhava_id", "<u8"), ("bytes_sent", "<u8"), ("bytes_rcvd", "<u8"), ("cli_pid", "<i4"), ("ser_pid", "<i4"),
    ("ser_conn_hash", "<u4"), ("ser_sock_inode", "<u4"), ("cli_comm", "S16"), ("ser_comm", "S16"), ("cli_cmdline_len", "<u2"),
    ("is_tcp_connect_event", "u1"), ("is_tcp_accept_event", "u1"), ("is_loopback_conn", "u1"), ("is_pre_existing", "u1"),
    ("notified_before", "u1"), ("padding_len", "u1")])
assert TCP_CONN_NOTIFY.itemsize == 280
assert TCP_CONN_NOTIFY.fields["ser_glob_id"][1] == 192 and TCP_CONN_NOTIFY.fields["bytes_sent"][1] == 208
assert TCP_CONN_NOTIFY.fields["cli_cmdline_len"][1] == 272 and TCP_CONN_NOTIFY.fields["padding_len"][1] == 279

# struct LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254): 88 fixed bytes (+ issue_string_len_ + padding_len_)
LISTENER_STATE_NOTIFY = np.dtype([
    ("glob_id", "<u8"), ("nqrys_5s", "<u4"), ("total_resp_5sec", "<u4"), ("nconns", "<u4"), ("nconns_active", "<u4"), ("ntasks", "<u4"),
    ("p95_5s_resp_ms", "<u4"), ("p95_5min_resp_ms", "<u4"), ("curr_kbytes_inbound", "<u4"), ("curr_kbytes_outbound", "<u4"),
    ("ser_errors", "<u4"), ("cli_errors", "<u4"), ("tasks_delay_usec", "<u4"), ("tasks_cpudelay_usec", "<u4"),
    ("tasks_blkiodelay_usec", "<u4"), ("tasks_user_cpu", "<u4"), ("tasks_sys_cpu", "<u4"), ("tasks_rss_mb", "<u4"),
    ("ntasks_issue", "<u2"), ("is_http_svc", "u1"), ("curr_state", "u1"), ("curr_issue", "u1"), ("issue_bit_hist", "u1"),
    ("high_resp_bit_hist", "u1"), ("last_issue_subsrc", "u1"), ("query_flags", "u1"), ("issue_string_len", "u1"), ("padding_len", "u1"),
    ("tail_pad", "u1")])
assert LISTENER_STATE_NOTIFY.itemsize == 88
assert LISTENER_STATE_NOTIFY.fields["curr_state"][1] == 79 and LISTENER_STATE_NOTIFY.fields["query_flags"][1] == 84

# struct tcp_ipv4_resp_event_t (common/gy_ebpf_kernel.h:106-111 + partha/gy_ebpf_kernel_struct.h:28-35): 24 bytes
RESP_EVENT = np.dtype([("saddr", "<u4"), ("daddr", "<u4"), ("netns", "<u4"), ("sport_be", ">u2"), ("dport_be", ">u2"),
                       ("lsndtime", "<u4"), ("lrcvtime", "<u4")])
assert RESP_EVENT.itemsize == 24
# tcp_ipv6_resp_event_t (common/gy_ebpf_kernel.h:113-118; ipv6_tuple_t partha/gy_ebpf_kernel_struct.h:37-44): 16-byte saddr / daddr, the rest
# laid out as in the IPv4 event
RESP_EVENT6 = np.dtype([("saddr", "u1", (16,)), ("daddr", "u1", (16,)), ("netns", "<u4"), ("sport_be", ">u2"), ("dport_be", ">u2"),
                        ("lsndtime", "<u4"), ("lrcvtime", "<u4")])
assert RESP_EVENT6.itemsize == 48

LISTEN_FLAG_DELETE = 0xC0


def splitmix64(x):
    """vectorised splitmix64 on uint64 arrays"""
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def machine_id(h):
    """machine_id = (splitmix(h), splitmix(h + 2^32)); returns 16 bytes (first, second little endian)"""
    with np.errstate(over="ignore"):
        a = splitmix64(np.uint64(h))
        b = splitmix64(np.uint64(h) + np.uint64(1 << 32))
    return int(a).to_bytes(8, "little") + int(b).to_bytes(8, "little")


def glob_id(h, s):
    """glob_id = splitmix(h * 2^20 + s) (never 0 / ~0)"""
    with np.errstate(over="ignore"):
        g = splitmix64(np.asarray(h, dtype=np.uint64) * np.uint64(1 << 20) + np.asarray(s, dtype=np.uint64))
    g = np.where((g == 0) | (g == np.uint64(0xFFFFFFFFFFFFFFFF)), np.uint64(1), g)
    return g


def listener_netns(h, s):
    return (0xF0000000 + 4 * np.asarray(h, dtype=np.int64) + np.asarray(s, dtype=np.int64) // 60000).astype(np.uint32)


def listener_port(s):
    return (1024 + np.asarray(s, dtype=np.int64) % 60000).astype(np.uint16)


def ip6_embedded_v4(a):
    """ip32_be_ of GY_IP_ADDR(unsigned __int128) for address bytes a[..., 16]: get_ipv6_type_flags (common/gy_common_inc.h:11040-11129) stores the
    IPv4 address an IPv6 address embeds -- 2002::/16 (bytes 2..5), ::ffff:a.b.c.d and 64:ff9b::/32 (bytes 12..15) -- into embedded_ipv4_"""
    a = np.asarray(a, dtype=np.uint8).reshape(-1, 16)
    u = lambda b: b[:, 0].astype(np.uint32) | (b[:, 1].astype(np.uint32) << 8) | (b[:, 2].astype(np.uint32) << 16) | (b[:, 3].astype(np.uint32) << 24)
    any_ = ~a.any(axis=1)
    loop = ~a[:, :15].any(axis=1) & (a[:, 15] == 1)
    g2 = (a[:, 0] & 0xF0) == 0x20
    six4 = g2 & (a[:, 0] == 0x20) & (a[:, 1] == 0x02)
    mapped = ~a[:, :10].any(axis=1) & (a[:, 10] == 0xFF) & (a[:, 11] == 0xFF)
    nat64 = (a[:, 0] == 0) & (a[:, 1] == 0x64) & (a[:, 2] == 0xFF) & (a[:, 3] == 0x9B)
    out = np.zeros(len(a), dtype=np.uint32)
    live = ~any_ & ~loop
    out = np.where(live & six4, u(a[:, 2:6]), out)
    out = np.where(live & ~g2 & mapped, u(a[:, 12:16]), out)
    out = np.where(live & ~g2 & ~mapped & nat64, u(a[:, 12:16]), out)
    return out


def set_ip_port(arr, ip32_be=None, ip128=None, port=0):
    """fills an IP_PORT field array the way GY_IP_ADDR::set_ip does (common/gy_common_inc.h:10673-10692)"""
    if ip128 is not None:
        arr["ip128"] = ip128
        arr["ip32_be"] = ip6_embedded_v4(np.asarray(arr["ip128"]))  # (embedded_ipv4_ shares its storage with ip32_be_, :10497-10500)
        arr["aftype"] = AF_INET6
    else:
        arr["ip128"] = 0
        arr["ip32_be"] = ip32_be
        arr["aftype"] = AF_INET
    arr["port"] = port


def pack_variable(fixed, tails):
    """packs fixed-size records + per-record tail bytes into one 8-byte aligned variable-stride batch the way the agent does
    (set_padding_len: pad every element to a multiple of 8).  `fixed` must already carry the tail length field.  Returns bytes."""
    out = bytearray()
    base = fixed.dtype.itemsize
    raw = fixed.tobytes()
    for i in range(len(fixed)):
        t = tails[i] if tails is not None else b""
        act = base + len(t)
        pad = (-act) % 8
        rec = bytearray(raw[i * base:(i + 1) * base])
        if fixed.dtype == TCP_CONN_NOTIFY:
            rec[272:274] = int(len(t)).to_bytes(2, "little")
            rec[279] = pad
        else:
            rec[85] = len(t)
            rec[86] = pad
        out += rec + t + b"\0" * pad
    return bytes(out)


def synth_tcp_conns(rng, n, hosts, svcs_per_host, dup_frac=0.2, v6_frac=0.0, close_frac=0.5, both_halves_frac=0.25, loopback_frac=0.05,
                    truth=None):
    """SURVEY 8d C2 flow stream as parthas report it: exactly n TCP_CONN_NOTIFY records that describe CONNECTIONS with a life cycle
    (server/gy_mconnhdlr.cc:9129-9341; common/gy_socket_stat.cc:1738-1787):
      * tuple: cli IP uniform in 10/8, cli port uniform 16000-65535, ser = (host IP, svc port); dup_frac of the connections reuse an
        earlier connection's tuple (reconnects: a new connection on a flow seen before);
      * life cycle: close_frac / 2 of the connections are short-lived (ONE record: closed, notified_before_ clear), close_frac / 2 are
        reported when they open AND when they close inside the batch (TWO records: open, then closed with notified_before_ set), the
        rest are still open (ONE open record); bytes ~ Pareto(1.2, 200) ride on the closing record only;
      * halves: every record above comes from the ACCEPTING partha (is_tcp_accept_event_); both_halves_frac of the connections are also
        reported by the CONNECTING partha (the same records again with is_tcp_connect_event_ only, ser_glob_id_ known for 70 % of
        them), loopback_frac are same-host connections reported once with both flags and is_loopback_conn_.
    hosts: array of host indices to draw the listeners from.  Records are shuffled.  `truth` (a dict) receives the per-CONNECTION arrays
    conn_host / conn_svc / conn_closed / conn_bytes_sent / conn_bytes_rcvd (what an exact connection table would hold)."""
    p_short, p_pair = close_frac / 2, close_frac / 2
    # draw n connections (always enough: a connection is at least one record), then keep the prefix whose records fit and fill the
    # remainder with one-record connections
    kind = rng.choice(3, n, p=[p_short, p_pair, 1 - p_short - p_pair])        # 0 short-lived, 1 open + close, 2 still open
    u = rng.random(n)
    half = np.where(u < loopback_frac, 2, np.where(u < loopback_frac + both_halves_frac, 1, 0))  # 0 accept only, 1 both halves, 2 loopback
    nrec_c = np.where(kind == 1, 2, 1) * np.where(half == 1, 2, 1)
    cum = np.cumsum(nrec_c)
    nc = int(np.searchsorted(cum, n, side="right"))
    fill = n - (int(cum[nc - 1]) if nc else 0)
    kind = np.concatenate([kind[:nc], np.zeros(fill, dtype=kind.dtype)])
    half = np.concatenate([half[:nc], np.zeros(fill, dtype=half.dtype)])
    nc += fill
    h = rng.choice(np.asarray(hosts), nc)
    s_ = rng.integers(0, svcs_per_host, nc)
    cli_ip = (0x0A000000 | rng.integers(0, 1 << 24, nc)).astype(">u4").view("<u4")
    cli_port = rng.integers(16000, 65536, nc).astype(np.uint16)
    ndup = int(nc * dup_frac)
    if ndup and nc > ndup:
        src = rng.integers(0, nc - ndup, ndup)
        idx = np.arange(nc - ndup, nc)
        for a in (h, s_, cli_ip, cli_port):
            a[idx] = a[src]
    closed_c = kind != 2
    sent_c = np.where(closed_c, ((rng.pareto(1.2, nc) + 1) * 200).astype(np.uint64), 0).astype(np.uint64)
    rcvd_c = np.where(closed_c, ((rng.pareto(1.2, nc) + 1) * 200).astype(np.uint64), 0).astype(np.uint64)
    task_c = (np.uint64(0x7A5C000000000000) + rng.integers(0, 64, nc).astype(np.uint64))
    resolved_c = rng.random(nc) < 0.7
    if truth is not None:
        truth.update(conn_host=h.copy(), conn_svc=s_.copy(), conn_closed=closed_c.copy(), conn_bytes_sent=sent_c.copy(), conn_bytes_rcvd=rcvd_c.copy())
    # expand into records: (connection, is the closing record, notified_before, is the connecting half)
    ci, isclose, notified, clihalf = [], [], [], []
    allc = np.arange(nc)
    for halfsel, cflag in ((allc, False), (allc[half == 1], True)):
        k = kind[halfsel]
        a = halfsel[k == 0]          # short-lived: one closing record, never notified before
        b = halfsel[k == 1]          # open record + closing record (notified before)
        c = halfsel[k == 2]          # still open
        for conns, cl, nb in ((a, True, False), (b, False, False), (b, True, True), (c, False, False)):
            ci.append(conns)
            isclose.append(np.full(len(conns), cl))
            notified.append(np.full(len(conns), nb))
            clihalf.append(np.full(len(conns), cflag))
    ci, isclose, notified, clihalf = (np.concatenate(x) for x in (ci, isclose, notified, clihalf))
    assert len(ci) == n, (len(ci), n)
    perm = rng.permutation(n)
    ci, isclose, notified, clihalf = ci[perm], isclose[perm], notified[perm], clihalf[perm]
    rec = np.zeros(n, dtype=TCP_CONN_NOTIFY)
    hh, ss = h[ci], s_[ci]
    ser_ip = (0x0A000000 | (hh.astype(np.int64) & 0xFFFFFF)).astype(">u4").view("<u4")
    ser_port = listener_port(ss)
    cli_arr = np.zeros(n, dtype=IP_PORT)  # (contiguous 32-byte elements: filling the fields of the 280-byte records in place is 5 x slower)
    ser_arr = np.zeros(n, dtype=IP_PORT)
    set_ip_port(cli_arr, ip32_be=cli_ip[ci], port=cli_port[ci])
    set_ip_port(ser_arr, ip32_be=ser_ip, port=ser_port)
    nv6 = int(nc * v6_frac)
    if nv6:
        c6 = rng.choice(nc, nv6, replace=False)        # per CONNECTION: every record of it carries the same v6 client address
        ip6_c = np.zeros((nc, 16), dtype=np.uint8)
        ip6_c[c6] = rng.integers(0, 256, (nv6, 16), dtype=np.uint8)
        ip6_c[c6, 0] = 0x20
        isv6 = np.zeros(nc, dtype=bool)
        isv6[c6] = True
        idx = np.nonzero(isv6[ci])[0]
        sub = cli_arr[idx]
        set_ip_port(sub, ip128=ip6_c[ci[idx]], port=cli_port[ci[idx]])
        cli_arr[idx] = sub
    rec["cli"] = cli_arr
    rec["nat_cli"] = cli_arr
    rec["ser"] = ser_arr
    rec["nat_ser"] = ser_arr
    gid = glob_id(hh, ss)
    rec["ser_glob_id"] = np.where(clihalf & ~resolved_c[ci], np.uint64(0), gid)
    rec["cli_task_aggr_id"] = np.where(clihalf | resolved_c[ci], task_c[ci], np.uint64(0))
    rec["tusec_start"] = 1_700_000_000_000_000 + ci.astype(np.uint64)
    rec["tusec_close"] = np.where(isclose, rec["tusec_start"] + 1000, 0)
    rec["bytes_sent"] = np.where(isclose, sent_c[ci], 0)
    rec["bytes_rcvd"] = np.where(isclose, rcvd_c[ci], 0)
    loop = half[ci] == 2
    rec["is_tcp_connect_event"] = (clihalf | loop).astype(np.uint8)
    rec["is_tcp_accept_event"] = (~clihalf).astype(np.uint8)
    rec["is_loopback_conn"] = loop.astype(np.uint8)
    rec["is_pre_existing"] = (rng.random(n) < 0.02).astype(np.uint8)
    rec["notified_before"] = notified.astype(np.uint8)
    rec["cli_comm"] = b"client"
    rec["ser_comm"] = b"server"
    return rec


def synth_listener_states(rng, host, svc_ids, delete_frac=0.0, bad_state_frac=0.0):
    """one LISTENER_STATE_NOTIFY per service of a host: nqrys_5s ~ Poisson(lambda_s), lambda_s lognormal(4, 2)"""
    n = len(svc_ids)
    rec = np.zeros(n, dtype=LISTENER_STATE_NOTIFY)
    rec["glob_id"] = glob_id(np.full(n, host), np.asarray(svc_ids))
    lam = np.minimum(rng.lognormal(4, 2, n), 1e6)
    rec["nqrys_5s"] = rng.poisson(lam)
    rec["total_resp_5sec"] = rec["nqrys_5s"] * rng.integers(1, 50, n)
    rec["nconns"] = rng.integers(0, 500, n)
    rec["nconns_active"] = rng.integers(0, 40, n) * (rng.random(n) < 0.7)
    rec["ntasks"] = rng.integers(1, 8, n)
    rec["p95_5s_resp_ms"] = rng.choice([1, 10, 30, 60, 100, 150, 200, 300], n)
    rec["p95_5min_resp_ms"] = rng.choice([1, 10, 30, 60, 100, 150, 200, 300], n)
    rec["curr_kbytes_inbound"] = rng.integers(0, 5000, n) * (rng.random(n) < 0.8)
    rec["curr_kbytes_outbound"] = rng.integers(0, 9000, n) * (rng.random(n) < 0.8)
    rec["ser_errors"] = rng.integers(0, 5, n)
    rec["tasks_delay_usec"] = rng.integers(0, 100000, n)
    rec["curr_state"] = rng.choice([0, 1, 2, 3, 4, 5], n, p=[0.2, 0.4, 0.2, 0.1, 0.07, 0.03])
    if bad_state_frac:
        rec["curr_state"] = np.where(rng.random(n) < bad_state_frac, 9, rec["curr_state"])
    if delete_frac:
        rec["query_flags"] = np.where(rng.random(n) < delete_frac, LISTEN_FLAG_DELETE, 0)
    return rec


# ---- COMM_HEADER / EVENT_NOTIFY framing (common/gy_comm_proto.h:336-420, :486-500)
PM_HDR_MAGIC = 0x05666605
COMM_EVENT_NOTIFY = 14
COMM_QUERY_CMD = 15
NOTIFY_LISTENER_STATE = 0x309
NOTIFY_TCP_CONN = 0x30C
NOTIFY_CPU_MEM_STATE = 0x30F


def frame_event_notify(subtype, nevents, payload, magic=PM_HDR_MAGIC, data_type=COMM_EVENT_NOTIFY):
    """one message as COMM_HEADER::set_type_len builds it: total_sz_ rounded up to 8, padding_sz_ = the difference"""
    act = 16 + 8 + len(payload)
    total = (act + 7) & ~7
    hdr = np.array([magic, total, data_type, total - act], dtype="<u4").tobytes()
    ev = np.array([subtype, nevents], dtype="<u4").tobytes()
    return hdr + ev + payload + b"\0" * (total - act)


# comm::ACTIVE_CONN_STATS (common/gy_comm_proto.h:2766-2783), 104 bytes fixed stride: one row per (listener, client task group) of a
# partha's 15-s active-connection report; flags byte @102: bit 0 cli_listener_proc_, bit 1 is_remote_listen_, bit 2 is_remote_cli_
ACTIVE_CONN_STATS = np.dtype([("listener_glob_id", "<u8"), ("cli_aggr_task_id", "<u8"), ("ser_comm", "S16"), ("cli_comm", "S16"),
                              ("remote_machine_id", "<u8", 2), ("remote_madhava_id", "<u8"), ("bytes_sent", "<u8"), ("bytes_received", "<u8"),
                              ("cli_delay_msec", "<u4"), ("ser_delay_msec", "<u4"), ("max_rtt_msec", "<f4"), ("active_conns", "<u2"), ("flags", "u1"),
                              ("tail_pad", "u1")])
assert ACTIVE_CONN_STATS.itemsize == 104
ACTIVE_FLAG_CLI_LISTENER_PROC, ACTIVE_FLAG_REMOTE_LISTEN, ACTIVE_FLAG_REMOTE_CLI = 1, 2, 4
MAX_NUM_ACTIVE_CONNS = 2048


def synth_active_conns(rng, n, host, svcs_per_host, ntasks=40, remote_listen_frac=0.25, unknown_frac=0.02):
    """n ACTIVE_CONN_STATS rows of one host: listener x client-task-group pairs (repeats allowed: several 15-s reports), bytes ~ Pareto"""
    rec = np.zeros(n, dtype=ACTIVE_CONN_STATS)
    s = rng.integers(0, svcs_per_host, n)
    rec["listener_glob_id"] = glob_id(np.full(n, host), s)
    unk = rng.random(n) < unknown_frac
    rec["listener_glob_id"] = np.where(unk, rng.integers(1, 1 << 62, n, dtype=np.uint64), rec["listener_glob_id"])
    rec["cli_aggr_task_id"] = splitmix64((np.uint64(host) << np.uint64(24)) + rng.integers(0, ntasks, n).astype(np.uint64) + np.uint64(0x7A5C))
    rec["ser_comm"] = b"svc"
    rec["cli_comm"] = b"cli"
    rec["bytes_sent"] = (200 * (1 + rng.pareto(1.2, n))).astype(np.uint64)
    rec["bytes_received"] = (200 * (1 + rng.pareto(1.2, n))).astype(np.uint64)
    rec["cli_delay_msec"] = rng.integers(0, 50, n)
    rec["ser_delay_msec"] = rng.integers(0, 50, n)
    rec["max_rtt_msec"] = rng.random(n).astype(np.float32) * 20
    rec["active_conns"] = rng.integers(1, 200, n)
    rec["flags"] = np.where(rng.random(n) < remote_listen_frac, ACTIVE_FLAG_REMOTE_LISTEN, 0) | np.where(rng.random(n) < 0.1, ACTIVE_FLAG_CLI_LISTENER_PROC, 0)
    return rec
