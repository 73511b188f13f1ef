#!/bin/bash
# round 6, call u: spread of the lean default line on one box (five runs), the default line once more exactly as the driver runs it, smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6u; mkdir -p $O
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5"
for i in 1 2 3 4 5; do
  timeout 300 python bench.py $LEAN --detail-out $O/run$i.json > $O/run$i.line 2> $O/run$i.err
  python - $O/run$i.json run$i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-6s %7.2f G %s  %8.3f ms " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05}, d.get("device_code"))
PY
done 2>&1 | tee $O/spread.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -c 2900 $O/bench_line.json; echo; cat $O/bench_time.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
