#!/bin/bash
# round 5, run g: coalesced half-event loads (libgysketch_x3.so: two 12-byte loads per lane + DPP swap) against the strided 8-byte loads
O=gpurun_out/r5g; mkdir -p $O
GYS_LIB=$PWD/gyeeta_amd/lib/libgysketch_x3.so python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -n 3
one() { # tag lib args...
  tag=$1; lib=$2; shift 2
  GYS_LIB=$PWD/gyeeta_amd/lib/$lib timeout 400 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5 --nbuf 3 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
one base_a libgysketch.so --td-pend-cap 1920
one x3_a libgysketch_x3.so --td-pend-cap 1920
one base_b libgysketch.so --td-pend-cap 1920
one x3_b libgysketch_x3.so --td-pend-cap 1920
one x3_c1 libgysketch_x3.so --hosts 1000 --svcs 100 --td-pend-cap 1920
one base_c1 libgysketch.so --hosts 1000 --svcs 100 --td-pend-cap 1920
