#!/bin/bash
# round 6, call p: where the event kernel's time goes at FULL size, by switching parts off (timing only, results wrong): -DGYS_RESP_DBG=1 build,
# GYS_DBG bits: 1 no flush, 2 no image, 4 no flow hash / HLL, 8 no all-service histogram adds, 16 no rank atomics (key counts), 32 hashes but no register traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6p; mkdir -p $O
run() { tag=$1; shift
	env GYS_LIB=$GRAFT_REPO_ROOT/gyeeta_amd/lib/libgysketch_dbg.so "$@" timeout 200 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3 --detail-out $O/$tag.json > $O/$tag.line 2> $O/$tag.err
	python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-28s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
( run dbg_build_all_on
  for d in 1 2 3 4 8 16 32 12 28 31 63; do run resp_dbg$d GYS_DBG=$d; done ) 2>&1 | tee $O/summary.txt
