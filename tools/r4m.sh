#!/bin/bash
# round 4, GPU call 13: C2, conn ingest: how many times the grid fills the CUs for a 2^22-record call -- 3 (default: 5504 records per
# workgroup), 6 (libgysketch_span6p: 2752), 2 (span2p: 8192), 1 (span1p: 16384)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_span6p libgysketch_span2p libgysketch_span1p libgysketch libgysketch_span6p libgysketch_span2p libgysketch_span1p; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_$lib.$RANDOM.json 2> $O/conn_$lib.err
done
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/conn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f G rec/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01}, d["checks"]["svc_nconn_equals_connection_table"], d["checks"]["tallies_consistent"])
    except Exception as e:
        print(f, "no result", e)
PY
