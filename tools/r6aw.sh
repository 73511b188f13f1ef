#!/bin/bash
# GPU call aw: is the occasional long C5 step (10.0 against 9.2 ms per step in 2 of 9 ten-step runs) tied to the class-1 routing?  C5 shape, ten-step runs, eight per routing, alternating
cd /root/repo; O=gpurun_out/r6aw; mkdir -p $O
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3 --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2"
for r in 1 2 3 4 5 6 7 8; do
 for v in bins general; do
  if [ $v = general ]; then export GYS_CLASS1_GENERAL=1; else unset GYS_CLASS1_GENERAL; fi
  timeout 200 python bench.py $LEAN --detail-out $O/c5_${v}_$r.json > /dev/null 2> $O/c5_${v}_$r.err
  python - $O/c5_${v}_$r.json "$v $r" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
print("%-12s %.3f ms per step, kernels %.3f ms" % (sys.argv[2], d["ms_per_step"], sum(v["ms"] for v in k.values())))
PY
 done
done | tee $O/runs.txt
