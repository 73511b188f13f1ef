#!/bin/bash
# usage: tools/gpu_check.sh <tag> [extra bench args]   -- GPU parity tests + bench at 2^26, 2^28 and the default 2^29 events/window; results under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --events $((1<<26)) --steps 10 --warmup 3 --prime-windows 12 "$@" > $O/bench_2p26.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --events $((1<<28)) --steps 10 --warmup 3 "$@" > $O/bench_2p28.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --steps 8 --warmup 2 "$@" > $O/bench_2p29.json 2>> $O/bench.err
cat $O/pytest.log
for f in $O/bench_2p26.json $O/bench_2p28.json $O/bench_2p29.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%.3f G ev/s  %.2f ms/step " % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v, 3) for k, v in {a: b["ms"] for a, b in d["roofline"]["kernels"].items() if b["ms"] > 0.05}.items()})
except Exception as e:
    print("bench failed:", e)
PY
done
tail -3 $O/bench.err
