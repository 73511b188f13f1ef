#!/bin/bash
# round 5, after r5q: the sub-runs' counter passes and timed-region traces once more (their step anchor is the window close's last kernel, now
# k_window_finish), then the default line that cites them.  The default workload's passes of r5q stay (same device code).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5r; mkdir -p $O
bash $R/tools/pmc_collect_workloads.sh r5r c2_conn c1 c5_zipf c3_levels > $O/workloads.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
(cd $R && time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err) > $O/bench_time.txt 2>&1
head -c 400 $O/bench_line.json; echo; cat $O/bench_time.txt; grep -A7 "^== " $O/workloads.txt | cut -c1-180
