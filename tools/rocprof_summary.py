#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd results .db (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`) into the plain-text
per-kernel summary committed under profiles/.
  usage: rocprof_summary.py results.db [out.txt] [--timed K --anchor SUBSTR]
With --timed K the summary covers only the TIMED region of a bench.py run: everything from the start of the K-th-from-last launch of
the kernel whose name contains SUBSTR (default "k_resp_host<16, false, false") -- bench.py's timed steps are the last K windows when
its untimed tails are switched off (--no-cpu-baseline --no-quantile-check --no-host-fed) -- so that set-up batches (de-phase pass,
priming windows, smaller generator launches) do not dilute the per-kernel averages.  Falls back to the whole-run view `top_kernels`
when the per-dispatch view is not there."""
import sqlite3
import sys

args = sys.argv[1:]
timed, anchor = 0, "k_resp_host<16, false, false"
if "--timed" in args:
    i = args.index("--timed")
    timed = int(args[i + 1])
    del args[i:i + 2]
if "--anchor" in args:
    i = args.index("--anchor")
    anchor = args[i + 1]
    del args[i:i + 2]
db = sqlite3.connect(args[0])
cur = db.cursor()
lines = None
if timed:
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        tcol = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
        ecol = "end" if "end" in cols else ("end_timestamp" if "end_timestamp" in cols else None)
        if not cols or tcol is None or ecol is None:
            raise RuntimeError("no per-dispatch view `kernels` with start/end columns: %s" % cols)
        rows = list(cur.execute(f'select name, "{tcol}", "{ecol}" from kernels order by "{tcol}"'))
        anchors = [r for r in rows if anchor in r[0]]
        if len(anchors) < timed:
            raise RuntimeError("only %d launches of the anchor kernel" % len(anchors))
        t0 = anchors[-timed][1]
        agg = {}
        for name, s, e in rows:
            if s < t0:
                continue
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1000.0
        tot = sum(v[1] for v in agg.values()) or 1.0
        lines = ["# rocprofv3 --kernel-trace summary of the TIMED region only: the last %d windows (from the start of the %d-th-from-last launch of '%s'); durations in microseconds" % (timed, timed, anchor),
                 "%-110s %10s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
        for name, (calls, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append("%-110s %10d %14.3f %12.3f %8.2f" % (name[:110], calls, us, us / calls, 100.0 * us / tot))
    except Exception as ex:  # noqa: BLE001 -- a schema we do not know: say so and give the whole-run view
        sys.stderr.write("rocprof_summary: timed-region view unavailable (%s); tables: %s\n" % (ex, [r[0] for r in cur.execute("select name from sqlite_master")][:40]))
        lines = None
if lines is None:
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# rocprofv3 --kernel-trace --stats summary, WHOLE run incl. set-up batches (durations in microseconds)",
             "%-110s %10s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows:
        lines.append("%-110s %10d %14.3f %12.3f %8.2f" % (name[:110], calls, tot, avg, pct))
text = "\n".join(lines) + "\n"
if len(args) > 1:
    open(args[1], "w").write(text)
else:
    sys.stdout.write(text)
