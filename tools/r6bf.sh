#!/bin/bash
# round 6 (final device code): the non-default shapes, one lean line each (tools/r5l.sh with the detail file): per-rank loads of an N = 2 / 4 / 8 run,
# 480-listener hosts, --levels 1 / 2, 25 x 4000-listener hosts, the connection stream with hosts mixed record by record, the IPv6 stream
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6bf; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
run() { name=$1; shift
	timeout 300 python bench.py "$@" $LEAN --detail-out $O/$name.json > $O/$name.line 2> $O/$name.err
	python - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-22s %7.2f G %s  %8.3f ms " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
( run default_c3
  run rank_of_2_5000_hosts --hosts 5000
  run rank_of_4_2500_hosts --hosts 2500
  run rank_of_8_1250_hosts --hosts 1250
  run hosts_480_listeners --hosts 20832 --svcs 480
  run levels_1 --levels 1
  run levels_2 --levels 2
  run c5_25x4000 --zipf-milli 1100 --hosts 25 --svcs 4000 --nbuf 2
  run c2_mixed --workload conn --conn-stream mixed
  run c2_messages --workload conn
  run ipv6_stream --ipv6 ) 2>&1 | tee $O/summary.txt
