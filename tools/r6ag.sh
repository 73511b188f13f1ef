#!/bin/bash
# GPU call ag (1024 threads, eight waves per SIMD): variants of the roll-up accumulate kernel (bytes requested ahead per wave, waves per workgroup), the global roll-up at 10^7 services each
cd /root/repo; O=gpurun_out/r6ag; mkdir -p $O
for r in 1 2; do
for lib in gyeeta_amd/lib/libgysketch.so $(ls gyeeta_amd/lib/libgysketch_rb_*.so); do
	tag=$(basename $lib .so)
	GYS_LIB=/root/repo/$lib timeout 300 python bench.py --no-cpu-baseline --no-host-fed --configs none --steps 3 --warmup 1 --detail-out $O/$tag.$r.json > $O/$tag.$r.line 2> $O/$tag.$r.err
	python - $O/$tag.$r.json $tag <<'PY'
import json, sys
try:
    q = json.load(open(sys.argv[1]))["quantile_scan"]
    print("%-28s rollup %.2f ms (first %.1f) kernels %s  p50 %s p99 %s" % (sys.argv[2], q["global_rollup_ms"], q["global_rollup_first_call_ms"], {k: round(v, 2) for k, v in q["global_rollup_kernels_ms"].items()}, q["global_rollup_quantiles_ms"]["p50"], q["global_rollup_quantiles_ms"]["p99"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done; done 2>&1 | tee $O/ab.txt
