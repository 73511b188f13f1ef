#!/bin/bash
# round 6, call q: the FLOOR of the event kernel (GYS_DBG=63: event loads + listener probe only) by load pattern, and each pattern with everything on
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6q; mkdir -p $O
run() { tag=$1; lib=$2; shift 2
	env GYS_LIB=$GRAFT_REPO_ROOT/gyeeta_amd/lib/$lib "$@" timeout 200 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3 --detail-out $O/$tag.json > $O/$tag.line 2> $O/$tag.err
	python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-28s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
( for v in base dma dmaahead x3 nosaddr pf; do run floor_$v libgysketch_f_$v.so GYS_DBG=63; run full_$v libgysketch_f_$v.so; done ) 2>&1 | tee $O/summary.txt
