#!/usr/bin/env python3
"""Side measurement (not the headline bench): TCP_CONN_NOTIFY ingest rate of config C2 (1 000 hosts x 100 services), device-resident
fixed-stride 280-byte records, HLL + 2 x Count-Min + exact per-service counters.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gyeeta_amd import capi, wire  # noqa: E402
from gyeeta_amd.engine import SketchEngine  # noqa: E402

nh, sp, n = 1000, 100, 1 << 22
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4   # records per window = reps x 2^22
eng = SketchEngine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
s = np.arange(sp)
for h in range(nh):
    mid = wire.machine_id(h)
    eng.register_host(mid, "c")
    eng.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s), wire.listener_netns(h, s), wire.listener_port(s))
rng = np.random.default_rng(1)
rec = wire.synth_tcp_conns(rng, n, np.arange(nh), sp, dup_frac=0.2)
d = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()).cuda()
off = torch.arange(0, n * 280, 280, dtype=torch.int32, device="cuda")
eng.profile(True)
for it in range(2):
    capi.check(eng.L.gys_ingest_tcp_conn_dev(eng.h, C.c_void_p(d.data_ptr()), C.c_void_p(off.data_ptr()), n))
eng.window_close()
eng.profile_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 5
for st in range(steps):
    for r in range(reps):
        capi.check(eng.L.gys_ingest_tcp_conn_dev(eng.h, C.c_void_p(d.data_ptr()), C.c_void_p(off.data_ptr()), n))
    eng.window_close()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = eng.profile_get()
print(json.dumps({"metric": "TCP_CONN_NOTIFY records/s (C2)", "value": steps * reps * n / dt, "records_per_window": reps * n,
                  "algorithmic_GBps": steps * reps * n * 280 / dt / 1e9, "conn_ms_per_launch": prof["conn"][0] / prof["conn"][1]}))
