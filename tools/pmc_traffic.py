#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
prescribes).  usage: pmc_traffic.py <fetch_dir> <write_dir> <events_per_launch> <service_keys> <skip_first_launches> <out.json>
Units: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, so it is doubled
(calibration in this repo: k_gen_resp writes exactly 24 B x events and WRITE_SIZE reports exactly that number of KiB)."""
import collections
import csv
import json
import os
import sys

NAMES = {"k_resp_host": "resp_host", "k_key_pass": "key_pass", "k_digest_merge": "digest_merge", "k_digest_huge": "digest_huge",
         "k_resp_pass1": "resp_pass1", "k_resp_scatter": "scatter"}


def per_kernel(root, counter, skip):
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(d, f))):
                    if r["Counter_Name"] != counter:
                        continue
                    for k, short in NAMES.items():
                        if k in r["Kernel_Name"]:
                            vals[short][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for short, disp in vals.items():
        ids = sorted(disp)
        # several instantiations of one kernel share a profile name: group consecutive dispatches per bench step
        per_step = collections.defaultdict(float)
        order = {d: i for i, d in enumerate(ids)}
        n_per_step = 2 if short == "digest_merge" else 1
        for d in ids:
            per_step[order[d] // n_per_step] += disp[d]
        steps = sorted(per_step)[skip:]
        if steps:
            out[short] = sum(per_step[s] for s in steps) / len(steps)
    return out


fetch_dir, write_dir, events, keys, skip, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
f = per_kernel(fetch_dir, "FETCH_SIZE", skip)
w = per_kernel(write_dir, "WRITE_SIZE", skip)
res = {"events_per_launch": events, "service_keys": keys,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; KiB -> bytes, FETCH_SIZE x2 (gfx950 correction)",
       "kernels": {k: {"fetch_bytes": int(f.get(k, 0) * 1024 * 2), "write_bytes": int(w.get(k, 0) * 1024)} for k in sorted(set(f) | set(w))}}
json.dump(res, open(outp, "w"), indent=1)
print(json.dumps(res))
