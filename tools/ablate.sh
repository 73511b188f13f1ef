#!/bin/bash
# timing experiments: steady-state bench with GYS_DBG_SKIP masks (results may be invalid; kernel times only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ablate
for m in "$@"; do
  GYS_DBG_SKIP=$m timeout 200 python bench.py --no-cpu-baseline --no-quantile-check --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('mask $m %.2f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernels_ms_avg'].items() if k in ('resp_host','key_pass','digest_merge')})"
done
