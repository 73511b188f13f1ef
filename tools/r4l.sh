#!/bin/bash
# round 4, GPU call 12: C2 with the conn ingest's workgroups spanning ~6144 records in a grid that is a multiple of the CU count (default:
# conn_span) against the fixed 8 x 768 of r4k (libgysketch_connr8) and 1536 records per workgroup (libgysketch_span1536), on the
# per-partha message stream and on the stream with hosts mixed record by record (the LDS table fills); then the conn parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
for s in messages mixed; do
for lib in libgysketch libgysketch_connr8 libgysketch_span1536 libgysketch libgysketch_connr8 libgysketch_span1536; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --conn-stream $s --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_${s}_$lib.$RANDOM.json 2> $O/conn_$lib.err
done; done
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/conn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f G rec/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01}, d["checks"]["svc_nconn_equals_connection_table"], d["checks"]["tallies_consistent"])
    except Exception as e:
        print(f, "no result", e)
PY
(time timeout 900 python -m pytest tests -m gpu -x -q -k "conn or round3 or configs" 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
