#!/bin/bash
# round 6, call t: what a one-probe listener table would save: GYS_DBG=64 (no third-and-later probes; timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6t; mkdir -p $O
run() { tag=$1; shift
	env GYS_LIB=$GRAFT_REPO_ROOT/gyeeta_amd/lib/libgysketch_dbg.so "$@" timeout 200 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5 --detail-out $O/$tag.json > $O/$tag.line 2> $O/$tag.err
	python - $O/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-20s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
PY
}
( run all_on; run no_third_probes GYS_DBG=64; run all_on_again; run no_third_probes_again GYS_DBG=64; run floor_no_third GYS_DBG=127; run floor GYS_DBG=63 ) 2>&1 | tee $O/summary.txt
