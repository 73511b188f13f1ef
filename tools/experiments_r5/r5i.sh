#!/bin/bash
# round 5, run i: --levels 1 with level 0 written by the fold pass against the separate k_level_roll pass (GYS_NO_FUSED_LAST)
O=gpurun_out/r5i; mkdir -p $O
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
GYS_FUSED_LAST=1 one fused --levels 1
one separate --levels 1
