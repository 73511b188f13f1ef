#!/bin/bash
# round 4, GPU call 18: k_conn_ingest with the per-workgroup HLL floor (default) against every record reading its register (libgysketch_nofloor), twice each;
# then the conn parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_nofloor libgysketch libgysketch_nofloor; do
	f=$O/conn_$lib.$RANDOM.json
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $f 2> $O/conn_$lib.err
	python - $f $lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01})
except Exception as e:
    print(sys.argv[2], "no result (the run's own checks fail on a switched-off build):", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee $O/summary.txt
(time timeout 900 python -m pytest tests -m gpu -x -q -k "conn or round3 or configs" 2>&1 | tail -6) 2>&1 | grep -v amdgpu | tee $O/pytest.log
