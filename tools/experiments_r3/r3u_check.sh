#!/bin/bash
# r3u: adaptive tile form (512 x 12 with two workgroups per CU when the tables are small): whole GPU suite + lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3u}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; shift; timeout 250 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
run default --steps 20 --warmup 5
run svcs480_auto --hosts 20832 --svcs 480 --steps 10 --warmup 3
GYS_TPT=16 run svcs480_tpt16 --hosts 20832 --svcs 480 --steps 10 --warmup 3 --no-quantile-check
run svcs100_auto --hosts 100000 --svcs 100 --steps 10 --warmup 3
GYS_TPT=16 run svcs100_tpt16 --hosts 100000 --svcs 100 --steps 10 --warmup 3 --no-quantile-check
run c1_shape_auto --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3 --no-quantile-check
GYS_TPT=16 run c1_shape_tpt16 --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3 --no-quantile-check
