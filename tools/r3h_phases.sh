#!/bin/bash
# phase timing of the two hot kernels by SKIPPING parts (timing only, results wrong): k_digest_bins through compile-time GYS_MB_SKIP
# builds (libgysketch_skip<mask>.so), k_resp_host through the GYS_DBG launch switches.  Quarter-size runs (2 500 hosts, 2^27 events per
# window: the per-key rates of the default line).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3h; mkdir -p $O
run() { tag=$1; lib=$2; shift 2
	env GYS_LIB=$R/gyeeta_amd/lib/$lib "$@" timeout 120 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --hosts 2500 --events 134217728 --steps 16 --warmup 3 > $O/$tag.json 2> $O/$tag.err
	python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run base libgysketch.so
for v in 1 2 4 8 16 32 63; do run mb_skip$v libgysketch_skip$v.so; done
for d in 1 2 3 4 8 16 28 31; do run resp_dbg$d libgysketch.so GYS_DBG=$d; done
