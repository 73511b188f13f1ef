#!/bin/bash
# round 5, run h: level 0 written by the fold pass (k_fold<LEVELS>): GPU suite, then the default workload with --levels 1 (lean) and without
O=gpurun_out/r5h; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
one() { tag=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 24 --warmup 6 --nbuf 2 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
one levels1 --levels 1
one levels0
one levels1_cap896 --levels 1 --td-pend-cap 0
