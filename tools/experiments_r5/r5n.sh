#!/bin/bash
# round 5, run n: k_digest_bins with the non-empty-bin compaction (default library) against the plain per-bin pass (libgysketch_nocompact.so)
O=gpurun_out/r5n; mkdir -p $O
python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_round5.py -m gpu -q -x 2>&1 | tail -n 3
one() { tag=$1; lib=$2; shift 2
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
one compact_a libgysketch.so
one plain_a libgysketch_nocompact.so
one compact_b libgysketch.so
one plain_b libgysketch_nocompact.so
one compact_cap896 libgysketch.so --td-pend-cap 0
one plain_cap896 libgysketch_nocompact.so --td-pend-cap 0
