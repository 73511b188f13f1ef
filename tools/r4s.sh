#!/bin/bash
# round 4, GPU call 19: k_conn_ingest without dependent global reads in the record walk (glob_id-keyed LDS table, parked HLL candidates: default) against r4r's kernel (libgysketch_nodefer), twice each,
# on the message stream and on the mixed stream; then the conn parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
for s in messages mixed; do for lib in libgysketch libgysketch_nodefer libgysketch libgysketch_nodefer; do
	f=$O/conn_$lib.$RANDOM.json
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --conn-stream $s --no-cpu-baseline --steps 30 --warmup 5 > $f 2> $O/conn_$lib.err
	python - $f $s:$lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01})
except Exception as e:
    print(sys.argv[2], "no result (the run's own checks fail on a switched-off build):", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done; done 2>&1 | tee $O/summary.txt
(time timeout 900 python -m pytest tests -m gpu -x -q -k "conn or round3 or configs" 2>&1 | tail -6) 2>&1 | grep -v amdgpu | tee $O/pytest.log
