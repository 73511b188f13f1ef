#!/bin/bash
# r3t: does a SECOND resident workgroup per CU pay at equal tile size?  480-listener hosts (their tables leave room for two 512-thread
# workgroups with 8192-event tiles): 1024 threads x 8 events (one workgroup per CU) against 512 threads x 16 events (two per CU)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3t}; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; lib=$2; shift 2; GYS_LIB=$R/gyeeta_amd/lib/$lib timeout 250 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
W="--hosts 5208 --svcs 480 --events 134217728 --steps 10 --warmup 3"
GYS_TPT=8 run one_wg_1024x8 libgysketch.so $W --no-quantile-check
run two_wg_512x16 libgysketch_t512.so $W
run one_wg_1024x16 libgysketch.so $W --no-quantile-check
GYS_TPT=12 run two_wg_512x12 libgysketch.so $W --no-quantile-check
