#!/bin/bash
# round 4, GPU call 17: where k_conn_ingest's 1.06 ms go -- builds with parts switched off (GYS_CONN_SKIP: results are wrong, times only):
# 8 record reads + staging only, 1 no flow hash / HLL, 16 hash but no HLL register access, 2 nothing after the HLL, 4 no LDS aggregation
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_skip8 libgysketch_skip1 libgysketch_skip16 libgysketch_skip2 libgysketch_skip4; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_$lib.json 2> $O/conn_$lib.err
	python - $O/conn_$lib.json $lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01})
except Exception as e:
    print(sys.argv[2], "no result (the run's own checks fail on a switched-off build):", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee $O/summary.txt
