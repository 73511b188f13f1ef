#!/bin/bash
# round-3 checkpoint: the whole GPU suite + the default line + the other shapes (levels 1 / 2, C5, many-listener hosts)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3d}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; shift; timeout 280 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
run default --steps 20 --warmup 5
run levels1 --levels 1 --steps 18 --warmup 4 --no-quantile-check
run levels2 --levels 2 --steps 18 --warmup 4 --no-quantile-check
run c5_50x2000 --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2
run c5_25x4000 --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 8 --warmup 2
