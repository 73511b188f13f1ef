#!/bin/bash
# round 5, last call: run-to-run spread of the default line and of the levels sub-run on one box (lean lines: no CPU legs, no sub-runs), and the GPU
# suite twice more on the final library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s %7.2f G %s  %8.3f ms  device_code %s " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"], d.get("device_code")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
( for i in 1 2 3 4 5; do timeout 200 python bench.py $LEAN > $O/default_$i.json 2> $O/default_$i.err; line $O/default_$i.json default_$i; done
  for i in 1 2 3; do timeout 200 python bench.py --levels 1 --nbuf 2 $LEAN > $O/levels_$i.json 2> $O/levels_$i.err; line $O/levels_$i.json levels1_$i; done
  for i in 1 2; do timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -n 1; done ) 2>&1 | tee $O/summary.txt
