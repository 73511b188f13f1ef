#!/bin/bash
# GPU call ax: the event kernel's tile form on the C1 shape (one host, 100 services, segments cut into 65 536-event parts, every key on a predicted run):
# default (512 x 12, two workgroups per CU) against GYS_TPT=16 (1024 x 16) and GYS_TPT=8 (1024 x 8); two rounds
cd /root/repo; O=gpurun_out/r6ax; mkdir -p $O
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3 --hosts 1 --svcs 100 --events 67108864 --nbuf 2"
for r in 1 2; do
 for t in 0 16 8; do
  if [ $t = 0 ]; then unset GYS_TPT; else export GYS_TPT=$t; fi
  timeout 200 python bench.py $LEAN --detail-out $O/c1_${t}_$r.json > /dev/null 2> $O/c1_${t}_$r.err
  python - $O/c1_${t}_$r.json "GYS_TPT=$t" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
print("%-12s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {a: round(v["ms"], 3) for a, v in k.items() if v["ms"] > 0.01})
PY
 done
done | tee $O/runs.txt
