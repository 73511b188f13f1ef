#!/bin/bash
# round 4, GPU call 21: does k_conn_ingest's work overlap its record reads?  default; without the prefetch (pf0); with the flow hash three
# times over (xh2 = -DGYS_CONN_EXTRA_HASH=2, a switch that existed for this call only: results wrong, time only) with and without prefetch; record reads + staging only (sk8) with and without prefetch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_pf0 libgysketch_xh2 libgysketch_xh2pf0 libgysketch_sk8 libgysketch_sk8pf0; do
	f=$O/conn_$lib.json
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $f 2> $O/conn_$lib.err
	python - $f $lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done 2>&1 | tee $O/summary.txt
