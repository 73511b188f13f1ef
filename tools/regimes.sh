#!/bin/bash
# per-rank load of the N-GPU runs emulated on one GPU: hosts/N hosts, the default 2^29 events per window
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for h in "$@"; do
  timeout 250 python bench.py --no-cpu-baseline --no-quantile-check --no-host-fed --hosts $h --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hosts $h: %.2f G ev/s %.2f ms' % (d['value']/1e9, d['ms_per_step']), {k: round(v,3) for k,v in {a: b['ms'] for a, b in d['roofline']['kernels'].items() if b['ms'] > 0.05}.items()})"
done
