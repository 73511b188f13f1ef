#!/usr/bin/env python3
"""FETCH_SIZE calibration for the event kernel's access pattern: a kernel that ONLY reads 2^28 resident 24-byte events
with 3 x 8-byte loads per thread at a 24-byte stride (k_read_events).  Run it under the counter pass and compare:

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/cal -o p --output-format csv -- python tools/calibrate_fetch.py
    python tools/pmc_kernels.py /tmp/cal k_read_events k_gen_resp

prints the known byte count; FETCH_SIZE (KiB) x 1024 x F = 24 B x events gives the factor F for this pattern."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gyeeta_amd import capi, wire  # noqa: E402
from gyeeta_amd.engine import SketchEngine  # noqa: E402

n = 1 << 28
eng = SketchEngine(max_hosts=4, max_services=64, enable_tdigest=False)
ev = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
eng.register_host(wire.machine_id(0), "c")
eng.gen_resp_events(ev.data_ptr(), n, 1, 0, 1, 8)
for _ in range(5):
    capi.check(eng.L.gys_debug_read_events_dev(eng.h, C.c_void_p(ev.data_ptr()), n))
eng.sync()
print("k_read_events: 5 launches x %d events x 24 B = %d bytes (%d KiB) per launch" % (n, n * 24, n * 24 // 1024))
eng.close()
