#!/bin/bash
# GPU call aj: the all-service quantile scan with the merge kernel's large-value list (1024 entries: 19.8 KB of LDS, 8 workgroups per CU) against its own (2048: 6 per CU)
cd /root/repo; O=gpurun_out/r6aj; mkdir -p $O
for r in 1 2; do
for lib in gyeeta_amd/lib/libgysketch.so gyeeta_amd/lib/libgysketch_scancap.so; do
	tag=$(basename $lib .so)
	GYS_LIB=/root/repo/$lib timeout 300 python bench.py --no-cpu-baseline --no-host-fed --configs none --steps 3 --warmup 1 --detail-out $O/$tag.$r.json > $O/$tag.$r.line 2> $O/$tag.$r.err
	python - $O/$tag.$r.json $tag <<'PY'
import json, sys
try:
    q = json.load(open(sys.argv[1]))["quantile_scan"]
    print("%-28s scan kernel %.2f ms  wall %.1f ms  p99 mean %.4f  rollup %.2f ms" % (sys.argv[2], q["kernel_ms"], q["wall_ms_incl_copy_to_host"], q["p99_mean_ms"], q["global_rollup_ms"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done; done 2>&1 | tee $O/ab.txt
