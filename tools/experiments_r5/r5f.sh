#!/bin/bash
# round 5, run f: non-temporal event loads (libgysketch_evnt.so) against the default library at td_pend_cap 1920; the 4096-value merge instance at 5 waves per SIMD (caps 2944 / 3968)
O=gpurun_out/r5f; mkdir -p $O
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
one cap1920_a libgysketch.so --td-pend-cap 1920
one cap1920_evnt libgysketch_evnt.so --td-pend-cap 1920
one cap1920_b libgysketch.so --td-pend-cap 1920
one cap1920_evnt_b libgysketch_evnt.so --td-pend-cap 1920
one cap2944 libgysketch.so --td-pend-cap 2944
one cap3968 libgysketch.so --td-pend-cap 3968
one cap1408 libgysketch.so --td-pend-cap 1408
