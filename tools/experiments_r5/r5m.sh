#!/bin/bash
# round 5, run m: predicted runs with the reservation counter (no compare-and-swap loop) + 256 values more room above the fast merge size: the per-rank
# loads of N = 2 / 4 / 8 and the 480-listener hosts again; then the scalar-base event loads (libgysketch_saddr.so) against the default library
O=gpurun_out/r5m; mkdir -p $O
one() { tag=$1; lib=$2; shift 2
  GYS_LIB=$PWD/gyeeta_amd/lib/$lib timeout 400 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3 --nbuf 3 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
one rank_of_2_5000 libgysketch.so --hosts 5000
one rank_of_4_2500 libgysketch.so --hosts 2500
one rank_of_8_1250 libgysketch.so --hosts 1250
one hosts_480 libgysketch.so --svcs 480
one base_a libgysketch.so
one saddr_a libgysketch_saddr.so
one base_b libgysketch.so
one saddr_b libgysketch_saddr.so
python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round5.py -m gpu -q -x 2>&1 | tail -n 3
GYS_LIB=$PWD/gyeeta_amd/lib/libgysketch_saddr.so python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -n 2
