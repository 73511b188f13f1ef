#!/bin/bash
# round 4, GPU call 25: where a large key's merge goes (k_huge_merge<512,512>, C5: 1.9 ms) -- a -DGYS_HUGE_TIMING build sums the shader
# clock per phase over the entries (experiment build: the counters line goes to stderr)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4z; mkdir -p $O; cd $R
GYS_LIB=$R/gyeeta_amd/lib/libgysketch_hugetime.so timeout 300 python bench.py --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 6 --warmup 2 --nbuf 2 --no-cpu-baseline --no-host-fed --no-quantile-check --configs none > $O/c5.json 2> $O/c5.err
grep GYS_HUGE_TIMING $O/c5.err | tail -3
python - $O/c5.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%.2f G ev/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
PY
GYS_LIB=$R/gyeeta_amd/lib/libgysketch_hugetime.so timeout 300 python bench.py --hosts 1 --svcs 100 --events 67108864 --steps 6 --warmup 2 --nbuf 2 --no-cpu-baseline --no-host-fed --no-quantile-check --configs none > $O/c1.json 2> $O/c1.err
grep GYS_HUGE_TIMING $O/c1.err | tail -2
