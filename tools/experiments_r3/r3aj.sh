#!/bin/bash
# r3aj: how does the merge kernel's throughput scale with the merges in flight per CU?  (extra dynamic LDS per workgroup: 8 -> 6 / 4 / 2 resident workgroups per CU)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3aj}; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
Q="--hosts 2500 --events 134217728 --steps 10 --warmup 3 --no-quantile-check"
for pad in 0 6500 20000 60000; do GYS_BINS_PAD=$pad timeout 200 python bench.py --no-cpu-baseline --no-host-fed $Q > $O/pad$pad.json 2> $O/pad$pad.err; line $O/pad$pad.json pad$pad; done
