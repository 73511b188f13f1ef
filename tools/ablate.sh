#!/bin/bash
# timing experiments: bench with GYS_DBG_SKIP masks (results invalid, kernel times only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ablate
for m in "$@"; do
  GYS_DBG_SKIP=$m timeout 200 python bench.py --no-cpu-baseline --events $((1<<28)) --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('mask $m', {k: round(v,3) for k,v in d['roofline']['kernels_ms_avg'].items() if k in ('resp_host','key_pass')})"
done
