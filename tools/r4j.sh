#!/bin/bash
# round 4, GPU call 10: C2 with the conn ingest at 12 waves in ONE workgroup per CU (default) against two workgroups of 6 waves, four of 3,
# and one of 16 waves with a 1024-entry aggregation table; then the conn parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_conn14t384 libgysketch_conn14t192 libgysketch_conn14t1024 libgysketch libgysketch_conn14t384 libgysketch_conn14t192 libgysketch_conn14t1024; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_$lib.$RANDOM.json 2> $O/conn_$lib.err
done
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/conn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f G rec/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01}, d.get("checks"))
    except Exception as e:
        print(f, "no result", e)
PY
(time timeout 900 python -m pytest tests -m gpu -x -q -k "conn" 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
