#!/bin/bash
# round 4, GPU call 3: all GPU parity tests with predicted runs on; default line (lean) for the histogram-cell change; C5 / C1 shapes with and
# without predicted runs (GYS_NO_PRESPILL=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
cat $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %.2f G/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5 > $O/default_lean.json 2> $O/default_lean.err; line $O/default_lean.json default
C5="--zipf-milli 1100 --hosts 50 --svcs 2000 --steps 20 --warmup 5 --nbuf 2"
C1="--hosts 1 --svcs 100 --events 67108864 --steps 20 --warmup 5 --nbuf 2"
timeout 300 python bench.py --sub c5_zipf $C5 > $O/c5_pre.json 2> $O/c5_pre.err; line $O/c5_pre.json c5_predicted_runs
GYS_NO_PRESPILL=1 timeout 300 python bench.py --sub c5_zipf $C5 > $O/c5_nopre.json 2> $O/c5_nopre.err; line $O/c5_nopre.json c5_no_prediction
timeout 300 python bench.py --sub c1 $C1 > $O/c1_pre.json 2> $O/c1_pre.err; line $O/c1_pre.json c1_predicted_runs
GYS_NO_PRESPILL=1 timeout 300 python bench.py --sub c1 $C1 --no-quantile-check > $O/c1_nopre.json 2> $O/c1_nopre.err; line $O/c1_nopre.json c1_no_prediction
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 10 --warmup 3 --nbuf 2 > $O/c5_4000.json 2> $O/c5_4000.err; line $O/c5_4000.json c5_25x4000
tail -2 $O/*.err | grep -v "^$" | grep -v amdgpu.ids | head -20
