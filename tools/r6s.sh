#!/bin/bash
# round 6, call s: Count-Min columns fixed at registration (k_cms_partial reads 2 bytes per service and row instead of hashing): the whole GPU suite, the lean default line, the IPv6 stream
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/tests.txt
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5"
for name in default ipv6; do
  extra=""; [ $name = ipv6 ] && extra="--ipv6"
  timeout 400 python bench.py $LEAN $extra --detail-out $O/$name.json > $O/$name.line 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-10s %7.2f G %s  %8.3f ms " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.004})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done 2>&1 | tee $O/lines.txt
