#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd results .db (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`) into the plain-text
per-kernel summary committed under profiles/.  usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
lines = ["# rocprofv3 --kernel-trace --stats summary (durations in microseconds)",
         "%-110s %10s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
for name, calls, tot, avg, pct in rows:
    lines.append("%-110s %10d %14.3f %12.3f %8.2f" % (name[:110], calls, tot, avg, pct))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
else:
    sys.stdout.write(text)
