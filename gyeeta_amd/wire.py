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


def set_ip_port(arr, ip32_be=None, ip128=None, port=0):
    """fills an IP_PORT field array the way GY_IP_ADDR::set_ip does (common/gy_common_inc.h:10673-10692)"""
    if ip128 is not None:
        arr["ip128"] = ip128
        arr["ip32_be"] = 0
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


def synth_tcp_conns(rng, n, hosts, svcs_per_host, dup_frac=0.2, v6_frac=0.0, close_frac=0.5):
    """SURVEY 8d C2 flow stream: cli IP uniform in 10/8, cli port uniform 16000-65535, ser = (host IP, svc port); dup_frac of the
    tuples repeat an earlier tuple (reconnects); bytes ~ Pareto(1.2, 200).  hosts: array of host indices to draw from."""
    rec = np.zeros(n, dtype=TCP_CONN_NOTIFY)
    h = rng.choice(np.asarray(hosts), n)
    s = rng.integers(0, svcs_per_host, n)
    cli_ip = (0x0A000000 | rng.integers(0, 1 << 24, n)).astype(">u4").view("<u4")
    cli_port = rng.integers(16000, 65536, n).astype(np.uint16)
    ndup = int(n * dup_frac)
    if ndup and n > ndup:
        src = rng.integers(0, n - ndup, ndup)
        idx = np.arange(n - ndup, n)
        for a in (h, s, cli_ip, cli_port):
            a[idx] = a[src]
    ser_ip = (0x0A000000 | (h.astype(np.int64) & 0xFFFFFF)).astype(">u4").view("<u4")
    ser_port = listener_port(s)
    for f in ("cli", "nat_cli"):
        set_ip_port(rec[f], ip32_be=cli_ip, port=cli_port)
    for f in ("ser", "nat_ser"):
        set_ip_port(rec[f], ip32_be=ser_ip, port=ser_port)
    nv6 = int(n * v6_frac)
    if nv6:
        idx = rng.choice(n, nv6, replace=False)
        ip6 = rng.integers(0, 256, (nv6, 16), dtype=np.uint8)
        ip6[:, 0] = 0x20
        sub = rec["nat_cli"][idx]
        set_ip_port(sub, ip128=ip6, port=cli_port[idx])
        rec["nat_cli"][idx] = sub
        rec["cli"][idx] = sub
    rec["ser_glob_id"] = glob_id(h, s)
    rec["tusec_start"] = 1_700_000_000_000_000 + np.arange(n, dtype=np.uint64)
    closed = rng.random(n) < close_frac
    rec["tusec_close"] = np.where(closed, rec["tusec_start"] + 1000, 0)
    rec["bytes_sent"] = np.where(closed, ((rng.pareto(1.2, n) + 1) * 200).astype(np.uint64), 0)
    rec["bytes_rcvd"] = np.where(closed, ((rng.pareto(1.2, n) + 1) * 200).astype(np.uint64), 0)
    rec["is_tcp_accept_event"] = 1
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
