#!/bin/bash
# round-3 final checkpoint: the whole GPU suite + every shape DESIGN.md section 9 quotes, on ONE tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3ah}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4) | tee $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    u = "G rec/s" if "conn" in d["roofline"].get("kernel", "") else "G ev/s"
    print("%-22s %6.2f %s %7.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, u, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; shift; timeout 280 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
run default --steps 20 --warmup 5
run h5000 --hosts 5000 --steps 10 --warmup 3 --no-quantile-check
run h2500 --hosts 2500 --steps 10 --warmup 3 --no-quantile-check
run h1250 --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
run svcs480 --hosts 20832 --svcs 480 --steps 10 --warmup 3
run levels1 --levels 1 --steps 12 --warmup 3 --no-quantile-check
run levels2 --levels 2 --steps 12 --warmup 3 --no-quantile-check
run c1_shape --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3
run c5_50x2000 --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2
run c5_25x4000 --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 8 --warmup 2 --no-quantile-check
run conn_messages --workload conn --steps 10 --warmup 3
run conn_mixed --workload conn --conn-stream mixed --steps 10 --warmup 3
