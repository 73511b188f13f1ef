#!/bin/bash
# round-3: event kernel without the tile-top barrier, HLL candidate queue, per-tile floor: whole GPU suite + lines (base library beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3o}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; lib=$2; shift 2; GYS_LIB=$R/gyeeta_amd/lib/$lib timeout 250 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
Q="--hosts 2500 --events 134217728 --steps 10 --warmup 3 --no-quantile-check"
run base_q libgysketch_base.so $Q
run new_q libgysketch.so $Q
run new_default libgysketch.so --steps 20 --warmup 5
run base_h1250 libgysketch_base.so --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
run new_h1250 libgysketch.so --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
run new_h5000 libgysketch.so --hosts 5000 --steps 10 --warmup 3 --no-quantile-check
run c5_50x2000 libgysketch.so --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2 --no-quantile-check
run c5_25x4000 libgysketch.so --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 8 --warmup 2 --no-quantile-check
run c1_shape libgysketch.so --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3 --no-quantile-check
run base_c1_shape libgysketch_base.so --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3 --no-quantile-check
