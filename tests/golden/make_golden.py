#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.json from the REFERENCE's own code (oracle/_ref/libgyref.so, built by
oracle/build_ref.sh from /root/reference).  Run in the build container only; the JSON is committed so the CPU and GPU
test suites need neither /root/reference nor oracle/_ref."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

R = o.ref()
assert R is not None, "run oracle/build_ref.sh first"
rng = np.random.default_rng(0x67796565)
g = {}

g["jhash2"] = []
for n in [1, 2, 3, 4, 5, 6, 7, 9, 10, 12]:
    for seed in (0xCEEDFEAD, 0x9E3779B9, 0xCEEDFEAE, 0):
        w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        g["jhash2"].append({"words": w.tolist(), "seed": seed, "hash": R.ref_jhash2(o.ptr(w, o.u32p), n, seed)})
g["jhash_bytes"] = []
for n in list(range(0, 30)) + [47, 64]:
    b = bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist())
    g["jhash_bytes"].append({"hex": b.hex(), "seed": 0xCEEDFEAD, "hash": R.ref_jhash(b, n, 0xCEEDFEAD)})
g["uint64_hash"] = [{"key": int(k), "hash": R.ref_get_uint64_hash(int(k))}
                    for k in [0, 1, 2**32, 2**64 - 1] + rng.integers(0, 2**63, 20, dtype=np.uint64).tolist()]
g["machine_id"] = [{"first": int(a), "second": int(b), "hash": R.ref_machine_id_hash(int(a), int(b))}
                   for a, b in rng.integers(0, 2**63, (16, 2), dtype=np.uint64).tolist()]


def ipval(v6):
    if v6:
        return bytes(rng.integers(0, 256, 16, dtype=np.uint8).tolist())
    return int(rng.integers(1, 2**32))


g["ip_port"] = []
for i in range(24):
    v6 = i % 3 == 2
    ip = ipval(v6) if i != 6 else 0  # 0.0.0.0 hashes as 16 zero bytes (get_as_inaddr)
    port = int(rng.integers(0, 65536))
    inode = int(rng.integers(1, 2**32))
    ign = i % 4 == 3
    ipb, f6 = o.ip_bytes(ip)
    g["ip_port"].append({"ip": ip.hex() if v6 else ip, "v6": int(v6), "port": port, "inode": inode, "ignore_ip": int(ign),
                         "hash": R.ref_ip_port_hash(ipb, f6, port, int(ign)),
                         "ns_hash": R.ref_ns_ip_port_hash(ipb, f6, port, inode, int(ign))})
g["pair_ip_port"] = []
for i in range(24):
    c6, s6 = i % 4 == 1, i % 4 >= 2
    cip, sip = ipval(c6), ipval(s6)
    cport, sport = int(rng.integers(0, 65536)), int(rng.integers(0, 65536))
    cb, cf = o.ip_bytes(cip)
    sb, sf = o.ip_bytes(sip)
    g["pair_ip_port"].append({"cip": cip.hex() if c6 else cip, "c6": int(c6), "cport": cport,
                              "sip": sip.hex() if s6 else sip, "s6": int(s6), "sport": sport,
                              "hash": R.ref_pair_ip_port_hash(cb, cf, cport, sb, sf, sport)})

g["hist"] = []
streams = {
    0: [np.minimum(np.floor(rng.lognormal(3.0, 1.5, 4000)), 1e6).astype(np.int64),
        np.array([0, 2, 2000, 1000000] + list(range(1000)), dtype=np.int64),
        np.array([-5, 0, 1, 15000, 15001, 2**40, -2**40], dtype=np.int64),
        np.array([], dtype=np.int64)],
    1: [(rng.pareto(1.0, 3000) * 50).astype(np.int64), np.array([-1, 0, 1, 5000000, 5000001, 2**31 + 7], dtype=np.int64)],
    2: [rng.poisson(300, 3000).astype(np.int64), np.array([150000, 150001, 2**32 + 3], dtype=np.int64)],
    3: [(rng.exponential(2000, 3000)).astype(np.int64)],
    4: [(rng.exponential(400, 3000)).astype(np.int64)],
    5: [rng.integers(-3, 300, 3000).astype(np.int64)],
    6: [rng.integers(-3, 3500, 3000).astype(np.int64)],
    7: [rng.integers(-5, 120, 2000).astype(np.int64),
        np.array([0, 2, 20, 90, 25, 35, 55, 65] + [i + i for i in range(100)], dtype=np.int64)],
    8: [rng.integers(0, 45, 500).astype(np.int64)],
    9: [rng.integers(-20, 5, 500).astype(np.int64)],
}
pcts = [1.0, 25.0, 50.0, 75.0, 90.0, 95.0, 99.0, 99.99, 100.0]
for kind, lst in streams.items():
    for vals in lst:
        h = R.ref_hist_new(kind)
        nb = R.ref_hist_nbuckets(h)
        v = np.ascontiguousarray(vals, dtype=np.int64)
        R.ref_hist_add_many(h, o.ptr(v, o.i64p), len(v))
        counts = np.zeros(nb, dtype=np.uint64)
        sums = np.zeros(nb, dtype=np.int64)
        total = C.c_uint64()
        maxv = C.c_int64()
        R.ref_hist_serialized(h, o.ptr(counts, o.u64p), o.ptr(sums, o.i64p), C.byref(total), C.byref(maxv))
        p = np.array(pcts, dtype=np.float32)
        pv = np.zeros(len(pcts), dtype=np.int64)
        ps = np.zeros(len(pcts), dtype=np.int64)
        pc = np.zeros(len(pcts), dtype=np.uint64)
        avg = C.c_float()
        t2 = C.c_uint64()
        m2 = C.c_int64()
        R.ref_hist_percentiles(h, o.ptr(p, o.f32p), len(pcts), o.ptr(pv, o.i64p), o.ptr(ps, o.i64p), o.ptr(pc, o.u64p),
                               C.byref(t2), C.byref(m2), C.byref(avg))
        g["hist"].append({"kind": kind, "values": v.tolist(), "nbuckets": nb, "counts": counts.tolist(), "sums": sums.tolist(),
                          "total": total.value, "max": maxv.value, "pcts": pcts, "pct_values": pv.tolist(),
                          "pct_sums": ps.tolist(), "pct_counts": pc.tolist(), "avg": float(avg.value),
                          "thresholds": [R.ref_hist_bucket_max_threshold(h, i) for i in range(nb + 1)]})
        R.ref_hist_free(h)

g["topn"] = []
for n, m in [(10, 100), (10, 5), (50, 1000), (3, 3)]:
    v = rng.integers(0, 200, m, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    k = R.ref_topn_u64(o.ptr(v, o.u64p), m, n, o.ptr(out, o.u64p))
    g["topn"].append({"n": n, "values": v.tolist(), "top": out[:k].tolist()})

g["sizes"] = {"GY_IP_ADDR": R.ref_sizeof(0), "IP_PORT": R.ref_sizeof(1), "PAIR_IP_PORT": R.ref_sizeof(2),
              "NS_IP_PORT": R.ref_sizeof(3), "HIST_SERIAL": R.ref_sizeof(4), "GY_HISTOGRAM_RESP": R.ref_sizeof(5),
              "HIST_DATA": R.ref_sizeof(6)}
ipb, _ = o.ip_bytes(0x0100007F)
buf = (C.c_uint8 * 32)()
R.ref_ip_port_bytes(ipb, 0, 8080, buf)
g["ip_port_layout_v4"] = bytes(buf).hex()
ip6 = bytes(range(1, 17))
ipb6, _ = o.ip_bytes(ip6)
R.ref_ip_port_bytes(ipb6, 1, 443, buf)
g["ip_port_layout_v6"] = bytes(buf).hex()

# thirdparty/SlabHistogramBucket.h (in tree): bucket numbering + percentile rule of the time levels (TIME_HISTOGRAM::get_stats).
# Drawn last so that the vectors above keep their values.
g["slab"] = {"nbuckets": R.ref_slab_num_buckets(), "bucket_idx": [], "percentile_idx": []}
for v in [-3, -1, 0, 1, 2, 10, 11, 30, 31, 59, 60, 61, 700, 701, 1000, 1001, 3000, 3001, 14999, 15000, 15001, 10**6, 2**40, -2**40] + \
        rng.integers(-10, 20000, 40).tolist():
    g["slab"]["bucket_idx"].append({"value": int(v), "idx": R.ref_slab_bucket_idx(int(v))})
for trial in range(60):
    kind = trial % 4
    if kind == 0:
        counts = rng.integers(0, 1000, 15)
    elif kind == 1:
        counts = rng.integers(0, 3, 15) * rng.integers(0, 10**6, 15)
    elif kind == 2:
        counts = np.zeros(15, dtype=np.int64)
        counts[rng.integers(0, 15)] = rng.integers(1, 10**9)
    else:
        counts = rng.integers(0, 2**40, 15)
    if trial == 59:
        counts = np.zeros(15, dtype=np.int64)  # empty histogram -> bucket 1
    counts = counts.astype(np.uint64)
    for pct in [0.0, 0.25, 0.5, 0.95, 0.99, 0.9999, 1.0] + rng.random(3).tolist():
        g["slab"]["percentile_idx"].append({"counts": counts.tolist(), "pct": float(pct),
                                            "idx": R.ref_slab_percentile_idx(o.ptr(counts, o.u64p), 15, float(pct))})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.json")
json.dump(g, open(out, "w"))
print("wrote", out, os.path.getsize(out), "bytes")
