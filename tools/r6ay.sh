#!/bin/bash
# GPU call ay: merge class 2 (4097 .. 16 384 values) through the streamed value-bin instance k_digest_bins<false,64> against the several-workgroup path (GYS_CLASS2_HUGE=1),
# C5 shape (ten-step runs, four per routing, alternating), C5 as 25 x 4000 and C1 once each; the parity tests that reach class 2 first
cd /root/repo; O=gpurun_out/r6ay; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "c1 or c5 or huge or zipf or spill or heavy or class or large" 2>&1 | tail -4 | tee $O/tests.txt
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
    print("%-22s %.2f G ev/s %.3f ms (kernels %.3f)" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], sum(v["ms"] for v in k.values())), {a: round(v["ms"], 3) for a, v in k.items() if v["ms"] > 0.04})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
for r in 1 2 3 4; do
 for v in bins huge; do
  if [ $v = huge ]; then export GYS_CLASS2_HUGE=1; else unset GYS_CLASS2_HUGE; fi
  timeout 200 python bench.py $LEAN --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --detail-out $O/c5_${v}_$r.json > /dev/null 2> $O/c5_${v}_$r.err; line $O/c5_${v}_$r.json "c5 class2=$v"
  if [ $r = 1 ]; then
   timeout 200 python bench.py $LEAN --zipf-milli 1100 --hosts 25 --svcs 4000 --nbuf 2 --detail-out $O/c5b_${v}.json > /dev/null 2> $O/c5b_${v}.err; line $O/c5b_${v}.json "c5 25x4000 class2=$v"
   timeout 200 python bench.py $LEAN --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --detail-out $O/c1_${v}.json > /dev/null 2> $O/c1_${v}.err; line $O/c1_${v}.json "c1 class2=$v"
  fi
 done
done | tee $O/ab.txt
