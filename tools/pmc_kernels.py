#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.  usage: pmc_kernels.py <dir-or-csv> [name-filter ...]"""
import collections
import csv
import os
import sys

paths = []
root = sys.argv[1]
if os.path.isdir(root):
    for d, _, fs in os.walk(root):
        paths += [os.path.join(d, f) for f in fs if f.endswith("counter_collection.csv")]
else:
    paths = [root]
filters = sys.argv[2:]
for path in paths:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if filters and not any(f in k for f in filters):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        n = len(disp[k])
        print("%-48s launches=%d  " % (k, n) + "  ".join("%s=%.4g" % (a, b / n) for a, b in sorted(v.items())))
