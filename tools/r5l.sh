#!/bin/bash
# round 5: the non-default shapes on the final kernels of the round (td_pend_cap 1920 by default; rank loads also at the 896-value buffer), one lean line each:
# per-rank loads of an N = 2 / 4 / 8 run (5000 / 2500 / 1250 hosts, 2^29 events), 480-listener hosts, --levels 1 / 2, 25 x 4000-listener
# hosts, the connection stream with hosts mixed record by record
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
run() { name=$1; shift
	timeout 300 python bench.py "$@" $LEAN > $O/$name.json 2> $O/$name.err
	python - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %7.2f G %s  %8.3f ms " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
( run rank_of_2_5000_hosts --hosts 5000
  run rank_of_4_2500_hosts --hosts 2500
  run rank_of_8_1250_hosts --hosts 1250
  run rank_of_2_5000_cap896 --hosts 5000 --td-pend-cap 0
  run rank_of_4_2500_cap896 --hosts 2500 --td-pend-cap 0
  run rank_of_8_1250_cap896 --hosts 1250 --td-pend-cap 0
  run hosts_480_listeners --hosts 10000 --svcs 480
  run levels_1 --levels 1
  run levels_2 --levels 2
  run c5_25x4000 --zipf-milli 1100 --hosts 25 --svcs 4000 --nbuf 2
  run c2_mixed --workload conn --conn-stream mixed ) 2>&1 | tee $O/summary.txt
