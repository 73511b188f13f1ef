#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3v}; mkdir -p $O
for lib in gyeeta_amd/lib/libgysketch.so $(ls gyeeta_amd/lib/libgysketch_cs*.so 2>/dev/null); do
	tag=$(basename $lib .so)
	GYS_LIB=$R/$lib timeout 120 python bench.py --workload conn --no-cpu-baseline --no-host-fed --steps 10 --warmup 3 > $O/$tag.json 2> $O/$tag.err
	python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-24s %.2f G rec/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
