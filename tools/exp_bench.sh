#!/bin/bash
# usage: tools/exp_bench.sh <tag> [pytest targets...]  -- quick A/B round on the GPU box: selected parity tests, then the default window
# overlapped and serialised (GYS_NO_OVERLAP=1), untimed legs skipped; prints one line per run
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
cd $R
mkdir -p gpurun_out
if [ $# -gt 0 ]; then timeout 500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4; fi
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%s %.3f G ev/s %.2f ms/step" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v, 2) for k, v in d["roofline"]["kernels_ms_avg"].items()})
except Exception as e:
    print(sys.argv[2], "bench failed:", e)
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check > gpurun_out/${T}_ov.json 2> gpurun_out/${T}.err
show gpurun_out/${T}_ov.json overlap
GYS_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --steps 6 --warmup 2 > gpurun_out/${T}_noov.json 2>> gpurun_out/${T}.err
show gpurun_out/${T}_noov.json serial
tail -2 gpurun_out/${T}.err
