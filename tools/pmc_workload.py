#!/usr/bin/env python3
"""HBM traffic per kernel and STEP of one bench.py workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as
MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe), merged into profiles/pmc_traffic.json under `workloads[<name>]`.

  usage: pmc_workload.py <fetch_dir> <write_dir> <name> <keep_last_steps> <json in/out> [units_per_step] [unit]

A step of every bench.py workload ends with the window close, whose last kernel is k_window_finish: step i = the dispatches after the
(i-1)-th and up to the i-th k_window_finish.  The timed steps are the last ones of a run whose untimed tails are switched off
(--sub / --no-cpu-baseline --no-quantile-check --no-host-fed): the last <keep_last_steps> steps are averaged.
Units (this environment's rocprofv3): both counters in KiB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B, so it is doubled
(calibrated in this repo on the event kernel's access pattern, profiles/r2n_calibrate_fetch.txt; WRITE_SIZE on k_gen_resp, which
writes exactly 24 B x events).  Kernel names are compacted: namespace and argument list dropped, template arguments kept
("k_resp_host<16,true,false,false>"); bench.py sums the kernels of a profile scope with scope_of()."""
import bisect
import collections
import csv
import json
import os
import re
import sys

ANCHOR = "k_window_finish"


def compact(kernel_name):
    """'void gys::k_resp_host<16u, true, false, false>(gys::RespHostP)' -> 'k_resp_host<16,true,false,false>'"""
    s = kernel_name
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):  # drop the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    s = s[:cut].strip()
    s = s.split(" ")[-1] if "<" not in s else s[s.rfind(" ", 0, s.index("<")) + 1:]
    s = s.replace("gys::", "").replace("(anonymous namespace)::", "")
    s = re.sub(r"(\d+)u\b", r"\1", s).replace(" ", "")
    return s


def scope_of(k):
    """the bench.py profile scope (gys_engine.hip ProfScope) a kernel runs under"""
    if k.startswith("k_resp_host<"):
        args = k[len("k_resp_host<"):-1].split(",")
        return "resp_spill" if len(args) >= 3 and args[2] in ("true", "1") else "resp_host"
    if k.startswith("k_digest_bins"):
        return "digest_merge"
    if k.startswith("k_digest_merge<"):
        return "digest_merge" if k.startswith("k_digest_merge<1024") else "digest_merge_big"
    if k.startswith("k_huge_") or k.startswith("k_digest_huge"):
        return "digest_huge"
    for pre, sc in (("k_key_finalize", "key_finalize"), ("k_fold", "fold"), ("k_conn_ingest", "conn"), ("k_lstate_", "lstate"),
                    ("k_actconn_ingest", "actconn"), ("k_level_", "level_roll"), ("k_resp_pass1", "resp_pass1"), ("k_resp_scatter", "scatter"),
                    ("k_key_append", "key_append"), ("k_scan_", "scan"), ("k_svc_filter", "svc_filter"), ("k_svc_aggr", "svc_aggr")):
        if k.startswith(pre):
            return sc
    if k.startswith(("k_conn_fold", "k_cms_", "k_window_prepare", "k_act_latch", "k_window_finish")):
        return "window_close"
    return "other"


def per_step(root, counter, keep):
    rows = []
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(d, f))):
                    if r["Counter_Name"] == counter and "gys::" in r["Kernel_Name"]:
                        rows.append((int(r["Dispatch_Id"]), compact(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    ends = [d for d, k, _ in rows if k.startswith(ANCHOR)]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d, k, v in rows:
        step = bisect.bisect_left(ends, d)  # dispatches up to and including the step's own k_window_finish
        if step < len(ends):
            per[k][step] += v
            launches[k][step] += 1
    last = list(range(max(0, len(ends) - keep), len(ends)))
    out = {}
    for k in per:
        out[k] = (sum(per[k].get(i, 0.0) for i in last) / max(len(last), 1), sum(launches[k].get(i, 0) for i in last) / max(len(last), 1))
    return out, len(ends)


def main():
    fetch_dir, write_dir, name, keep, path = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    units = int(sys.argv[6]) if len(sys.argv) > 6 else None
    unit = sys.argv[7] if len(sys.argv) > 7 else None
    f, nf = per_step(fetch_dir, "FETCH_SIZE", keep)
    w, nw = per_step(write_dir, "WRITE_SIZE", keep)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        from gyeeta_amd.build import build_commit, sources_sha, device_code_sha
        commit, ksha, dsha = build_commit(), sources_sha(), device_code_sha()
    except Exception:
        commit = ksha = dsha = None
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fb = int(f.get(k, (0.0, 0))[0] * 1024 * 2)
        wb = int(w.get(k, (0.0, 0))[0] * 1024)
        if fb + wb == 0:
            continue
        kernels[k] = {"scope": scope_of(k), "launches_per_step": f.get(k, w.get(k))[1], "fetch_bytes": fb, "write_bytes": wb}
    ent = {"units_per_step": units, "unit": unit, "steps_averaged": keep, "steps_seen": [nf, nw], "source_commit": commit, "source_kernels": ksha,
           "device_code": dsha, "kernels": kernels}
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    doc.setdefault("workloads", {})[name] = ent
    doc["workloads_note"] = ("per workload: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on the bench.py sub-run of that name "
                             "(tools/pmc_collect_workloads.sh); bytes per kernel and step = average over the last steps_averaged steps, a step ends with "
                             "k_window_finish; KiB -> bytes, FETCH_SIZE x2 (gfx950 correction)")
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps({name: ent}))


if __name__ == "__main__":
    main()
