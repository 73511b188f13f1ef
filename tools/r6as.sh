#!/bin/bash
# GPU call as: merge class 1 (2049 .. 4096 values) through k_digest_bins<false,16> against round 5's routing (GYS_CLASS1_GENERAL=1: the general kernel), C5 and C1 shapes and the
# default workload at a quarter of its size, two rounds; then the parity tests that exercise class 1 (C5 / Zipf / spill / large keys) on the new routing
cd /root/repo; O=gpurun_out/r6as; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-22s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.04})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
for r in 1 2; do
 for v in bins general; do
  if [ $v = general ]; then export GYS_CLASS1_GENERAL=1; else unset GYS_CLASS1_GENERAL; fi
  timeout 200 python bench.py $LEAN --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --detail-out $O/c5_${v}_$r.json > /dev/null 2> $O/c5_${v}_$r.err; line $O/c5_${v}_$r.json "c5 class1=$v"
  timeout 200 python bench.py $LEAN --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --detail-out $O/c1_${v}_$r.json > /dev/null 2> $O/c1_${v}_$r.err; line $O/c1_${v}_$r.json "c1 class1=$v"
  timeout 200 python bench.py $LEAN --hosts 2500 --events 134217728 --detail-out $O/q_${v}_$r.json > /dev/null 2> $O/q_${v}_$r.err; line $O/q_${v}_$r.json "quarter class1=$v"
 done
done | tee $O/ab.txt
unset GYS_CLASS1_GENERAL
timeout 900 python -m pytest tests -m gpu -x -q -k "c1 or c5 or huge or zipf or spill or heavy or class or large" 2>&1 | tail -4 | tee $O/tests.txt
