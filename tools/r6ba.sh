#!/bin/bash
# GPU call ba: the default workload (full size) with and without the class-2 routing (the default has no class-2 keys: two more launches and the hand-over pass through the 16 384-value general instance), two rounds
cd /root/repo; O=gpurun_out/r6ba; mkdir -p $O
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5"
for r in 1 2; do
 for v in bins huge; do
  if [ $v = huge ]; then export GYS_CLASS2_HUGE=1; else unset GYS_CLASS2_HUGE; fi
  timeout 300 python bench.py $LEAN --detail-out $O/d_${v}_$r.json > /dev/null 2> $O/d_${v}_$r.err
  python - $O/d_${v}_$r.json "default class2=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
print("%-22s %.2f G ev/s %.3f ms (kernels %.3f)" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], sum(v["ms"] for v in k.values())), {a: round(v["ms"], 3) for a, v in k.items() if v["ms"] > 0.01})
PY
 done
done | tee $O/ab.txt
