#!/bin/bash
# round 4, GPU call 22: k_conn_ingest with and without the prefetch of the next round, the kernel of r4n (slot-keyed table, per-record HLL and lookup reads: oldpf1 / oldpf0)
# against the current one (glob_id-keyed table, parked HLL candidates: default = prefetch, pf0 = none), twice each
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
for lib in libgysketch_pf0 libgysketch_oldpf1 libgysketch_oldpf0 libgysketch_pf0 libgysketch_oldpf1 libgysketch_oldpf0; do
	f=$O/conn_$lib.$RANDOM.json
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
