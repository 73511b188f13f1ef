#!/bin/bash
# r3ak: the two-wave merge kernel (k_digest_bins2, GYS_BINS2=1) against the four-wave one: parity tests with it switched on + lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3ak}; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; shift; timeout 250 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
(GYS_BINS2=1 timeout 600 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | tail -3) | tee $O/pytest.log
GYS_BINS2=0 run bins_default --steps 20 --warmup 5 --no-quantile-check
GYS_BINS2=1 run bins2_default --steps 20 --warmup 5
GYS_BINS2=1 run bins2_h1250 --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
GYS_BINS2=0 run bins_h1250 --hosts 1250 --steps 10 --warmup 3 --no-quantile-check
