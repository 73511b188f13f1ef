#!/bin/bash
# round 5, run e: full GPU suite, then the default line (lean) at td_pend_cap 896 / 1920 / 3968
O=gpurun_out/r5e; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
for cap in 0 1920 3968; do
  timeout 400 python bench.py --no-cpu-baseline --no-host-fed --configs none --steps 20 --warmup 5 --nbuf 3 --td-pend-cap $cap > $O/bench_cap$cap.json 2> $O/bench_cap$cap.err
  python - $O/bench_cap$cap.json $cap <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("cap", sys.argv[2], "%.2f G ev/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), "parity", d.get("parity_ok"), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02},
          "qerr", {k: round(v, 5) for k, v in (d.get("quantile_error") or {}).items() if k.endswith("max")}, "merges/step", d.get("td_merges_per_step"))
except Exception as e:
    print("cap", sys.argv[2], "failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
