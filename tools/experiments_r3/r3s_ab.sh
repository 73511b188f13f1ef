#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3s}; mkdir -p $O
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
(timeout 600 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4) | tee $O/pytest.log
run head_q libgysketch_head.so $Q
run new_q libgysketch.so $Q
run head_default libgysketch_head.so --steps 20 --warmup 5 --no-quantile-check
run new_default libgysketch.so --steps 20 --warmup 5
run head_h1250 libgysketch_head.so --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
run new_h1250 libgysketch.so --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
