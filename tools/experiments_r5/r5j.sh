#!/bin/bash
# round 5, run j: full GPU suite, then the default line as the driver runs it (sub-runs, CPU legs at 10^7 and 10^6 keys), timed
O=gpurun_out/r5j; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt
cat $O/bench_time.txt; tail -n 5 $O/bench.err
python - $O/bench_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%.2f G ev/s %.3f ms frac %.3f" % (d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
print("parity", d.get("parity_ok"), d.get("quantile_error"))
print("scan", d.get("quantile_scan"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:1500])
for k, v in d.get("configs", {}).items(): print(" cfg", k, v.get("value"), v.get("ms_per_step"), v.get("parity_ok"), v.get("wall_s"), v.get("error"))
print("host_fed", json.dumps(d.get("host_fed"))[:600])
PY
