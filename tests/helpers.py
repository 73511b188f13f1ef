"""shared helpers for the parity tests: register the same hosts / listeners in the GPU engine and in the CPU oracle engine"""
import numpy as np

from gyeeta_amd import wire


def register_world(eng, orc, hosts, svcs_per_host, cluster_of=None):
    """hosts: iterable of synthetic host indices.  Returns {host_index: (machine_id, host_slot)} and glob_id matrix [len(hosts)][svcs]."""
    info = {}
    gids = {}
    for h in hosts:
        mid = wire.machine_id(h)
        slot = eng.register_host(mid, (cluster_of or (lambda x: "cluster%d" % (x % 3)))(h)) if eng is not None else len(info)
        s = np.arange(svcs_per_host)
        g = wire.glob_id(np.full(svcs_per_host, h), s)
        ns = wire.listener_netns(h, s)
        pt = wire.listener_port(s)
        if eng is not None:
            eng.register_listeners_np(mid, g, ns, pt)
        if orc is not None:
            for i in range(svcs_per_host):
                orc.register(slot, int(g[i]), int(ns[i]), int(pt[i]))
        info[h] = (mid, slot)
        gids[h] = g
    return info, gids


def make_resp_events(rng, h, n, svcs_per_host, lat_mu=3.0, lat_sigma=1.5, bad_frac=0.02, unknown_frac=0.02, zero_ip_frac=0.01):
    """numpy RESP_EVENT batch for synthetic host h with the edge cases the reference filters: negative / > 1e6 latencies
    (common/gy_socket_stat.cc:1521-1524), events for ports without a listener, and 0.0.0.0 client addresses (get_as_inaddr quirk)."""
    ev = np.zeros(n, dtype=wire.RESP_EVENT)
    s = rng.integers(0, svcs_per_host, n)
    ev["saddr"] = int.from_bytes((0x0A000000 | (h & 0xFFFFFF)).to_bytes(4, "big"), "little")  # 10.x.y.z as ip32_be
    ev["daddr"] = (0x0A000000 | rng.integers(0, 1 << 24, n)).astype(">u4").view("<u4")
    ev["netns"] = wire.listener_netns(h, s)
    ev["sport_be"] = wire.listener_port(s)
    ev["dport_be"] = rng.integers(16000, 65536, n)
    lat = np.minimum(np.floor(rng.lognormal(lat_mu, lat_sigma, n)), 1e6).astype(np.uint32)
    lrcv = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    bad = rng.random(n) < bad_frac
    lat = np.where(bad & (rng.random(n) < 0.5), np.uint32(1000001) + rng.integers(0, 1000, n).astype(np.uint32), lat)
    neg = bad & (lat <= 1000000)
    with np.errstate(over="ignore"):
        ev["lsndtime"] = np.where(neg, lrcv - np.uint32(5), lrcv + lat)  # negative response -> dropped
    ev["lrcvtime"] = lrcv
    unk = rng.random(n) < unknown_frac
    ev["sport_be"] = np.where(unk, 999, ev["sport_be"])
    z = rng.random(n) < zero_ip_frac
    ev["daddr"] = np.where(z, 0, ev["daddr"])
    return ev


def assert_hist_equal(gpu_hist, orc_hist, nsvc):
    """both [nsvc][16][2] int64: buckets 0..14 {count,sum}, [15] = {total_count, max_val_seen}"""
    g = np.asarray(gpu_hist)[:nsvc]
    o = np.asarray(orc_hist)[:nsvc]
    bad = np.argwhere(g != o)
    assert bad.size == 0, f"histogram mismatch at {bad[:5].tolist()}: gpu {g[tuple(bad[0])]} oracle {o[tuple(bad[0])]}"


def concat_events(parts):
    """byte-level concatenation of RESP_EVENT arrays.  (np.concatenate would normalise the big-endian port fields to native byte
    order, i.e. silently rewrite the wire bytes.)"""
    raw = b"".join(p.tobytes() for p in parts)
    return np.frombuffer(raw, dtype=wire.RESP_EVENT)
