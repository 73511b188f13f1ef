#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3n}; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; lib=$2; shift 2; GYS_LIB=$R/gyeeta_amd/lib/$lib timeout 200 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
Q="--hosts 2500 --events 134217728 --steps 10 --warmup 3 --no-quantile-check"
(timeout 600 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round3.py -x -q 2>&1 | tail -5) | tee $O/pytest.log
run base_q libgysketch_base.so $Q
run new_q libgysketch.so $Q
run pf_q libgysketch_pf.so $Q
GYS_TPT=8 run new_tpt8_q libgysketch.so $Q
GYS_TPT=8 run pf_tpt8_q libgysketch_pf.so $Q
GYS_TBL_SPARSE=0 GYS_TPT=12 run new_tpt12_q libgysketch.so $Q
GYS_TBL_SPARSE=0 GYS_TPT=12 run pf_tpt12_q libgysketch_pf.so $Q
for d in 0 32; do GYS_DBG=$d run dbg${d}_q libgysketch_dbg.so $Q; done
run new_default libgysketch.so --steps 20 --warmup 5
run pf_default libgysketch_pf.so --steps 20 --warmup 5 --no-quantile-check
