#!/bin/bash
# round 4, GPU call 1: parity tests on the merged tree, A/B of the merge's paired-search lead, default line incl. sub-runs, C2 counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
tools/ab_libs.sh bench $O/ab_quarter --hosts 2500 --events 134217728 --steps 8 --warmup 2 > $O/ab_quarter.txt 2>&1
for lib in libgysketch libgysketch_g1 libgysketch_g4; do GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5 > $O/full_$lib.json 2> $O/full_$lib.err; done
(time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err) > $O/bench_time.txt 2>&1
bash tools/pmc_collect_workloads.sh r4a c2_conn > $O/pmc_c2.txt 2>&1
cat $O/pytest.log; cat $O/ab_quarter.txt
for f in $O/full_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "%.2f G ev/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
done
cat $O/bench_time.txt; tail -3 $O/bench_line.err; head -c 300 $O/bench_line.json; echo; tail -12 $O/pmc_c2.txt
