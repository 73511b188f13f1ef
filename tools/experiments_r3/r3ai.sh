#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3ai}; mkdir -p $O
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
(timeout 600 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round3.py -x -q 2>&1 | tail -3) | tee $O/pytest.log
C5="--zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2"
run head_c5 libgysketch_head.so $C5 --no-quantile-check
run new_c5 libgysketch.so $C5
run head_c1 libgysketch_head.so --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3 --no-quantile-check
run new_c1 libgysketch.so --hosts 1 --svcs 100 --events 67108864 --steps 10 --warmup 3
run new_c5_25x4000 libgysketch.so --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 8 --warmup 2 --no-quantile-check
