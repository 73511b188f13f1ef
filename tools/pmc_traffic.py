#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
prescribes).  usage: pmc_traffic.py <fetch_dir> <write_dir> <events_per_launch> <service_keys> <keep_last_steps> <out.json>
Units: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, so it is doubled
(calibration in this repo: k_gen_resp writes exactly 24 B x events and WRITE_SIZE reports exactly that number of KiB)."""
import collections
import csv
import json
import os
import sys

NAMES = {"k_resp_host": "resp_host", "k_key_finalize": "key_finalize", "k_fold": "fold", "k_digest_merge": "digest_merge", "k_digest_bins": "digest_merge",
         "k_cms_partial": "window_prepare", "k_cms_reduce": "window_prepare",
         "k_digest_huge": "digest_huge", "k_resp_pass1": "resp_pass1", "k_resp_scatter": "scatter", "k_window_prepare": "window_prepare"}


def short_name(kernel, k, short):
    """the second (SPILL) pass of k_resp_host is its own stage: template arguments <TPT, SHARED, SPILL, SVCHLL>"""
    if k == "k_resp_host":
        i = kernel.find("k_resp_host<")
        if i >= 0:
            args = [a.strip() for a in kernel[i + len("k_resp_host<"):kernel.find(">", i)].split(",")]
            if len(args) >= 3 and args[2] in ("true", "1"):
                return "resp_spill"
    return short


def per_kernel(root, counter, skip):
    """bytes-counter per kernel and STEP: a bench step starts at a k_resp_host dispatch; every dispatch of the other pipeline kernels
    up to the next k_resp_host belongs to it (key ranges and merge size classes are several launches per step)"""
    rows = []
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(d, f))):
                    if r["Counter_Name"] != counter:
                        continue
                    for k, short in NAMES.items():
                        if k in r["Kernel_Name"]:
                            rows.append((int(r["Dispatch_Id"]), short_name(r["Kernel_Name"], k, short), float(r["Counter_Value"])))
    rows.sort()
    starts = sorted({d for d, short, _ in rows if short == "resp_host"})
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    import bisect
    for d, short, v in rows:
        step = bisect.bisect_right(starts, d) - 1
        if step >= 0:
            per[short][step] += v
    out = {}
    for short, steps in per.items():
        # only full-size steps: the set-up passes before the first timed window use other batch sizes; keep the LAST (nsteps - skip)
        ids = sorted(steps)[-skip:]  # the timed windows are the last ones of the run
        if ids:
            out[short] = sum(steps[i] for i in ids) / len(ids)
    return out


fetch_dir, write_dir, events, keys, skip, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
f = per_kernel(fetch_dir, "FETCH_SIZE", skip)
w = per_kernel(write_dir, "WRITE_SIZE", skip)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from gyeeta_amd.build import build_commit, sources_sha, device_code_sha
    commit = build_commit()
    ksha = sources_sha()
    dsha = device_code_sha()
except Exception:
    commit = ksha = dsha = None
res = {"events_per_launch": events, "service_keys": keys, "source_commit": commit, "source_kernels": ksha, "device_code": dsha,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; KiB -> bytes, FETCH_SIZE x2 (gfx950 correction)",
       "kernels": {k: {"fetch_bytes": int(f.get(k, 0) * 1024 * 2), "write_bytes": int(w.get(k, 0) * 1024)} for k in sorted(set(f) | set(w))}}
json.dump(res, open(outp, "w"), indent=1)
print(json.dumps(res))
