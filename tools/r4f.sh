#!/bin/bash
# round 4, GPU call 6: the event kernel as 512 threads x 32 events with the next group's events prefetched into registers (GYS_TPT=32) against
# the 1024 x 16 form: parity tests under the variant, then A/B at full and quarter size on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
(GYS_TPT=32 timeout 600 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py tests/test_gpu_round4.py -m gpu -x -q -s 2>&1 | tail -8) > $O/pytest_tpt32.log 2>&1; grep -v amdgpu $O/pytest_tpt32.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %.2f G/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none"
for rep in 1 2; do
  timeout 300 python bench.py $LEAN --steps 20 --warmup 5 > $O/full_16_$rep.json 2> $O/err.txt; line $O/full_16_$rep.json full_1024x16
  GYS_TPT=32 timeout 300 python bench.py $LEAN --steps 20 --warmup 5 > $O/full_32_$rep.json 2>> $O/err.txt; line $O/full_32_$rep.json full_512x32_prefetch
done
timeout 300 python bench.py $LEAN --hosts 2500 --events 134217728 --steps 12 --warmup 3 > $O/q_16.json 2>> $O/err.txt; line $O/q_16.json quarter_1024x16
GYS_TPT=32 timeout 300 python bench.py $LEAN --hosts 2500 --events 134217728 --steps 12 --warmup 3 > $O/q_32.json 2>> $O/err.txt; line $O/q_32.json quarter_512x32_prefetch
GYS_TPT=32 timeout 300 python bench.py --no-cpu-baseline --no-host-fed --configs none --steps 10 --warmup 3 > $O/full_32_parity.json 2>> $O/err.txt; python -c "
import json; d=json.loads(open('$O/full_32_parity.json').read().strip().splitlines()[-1]); print('GYS_TPT=32 line with the quantile check: parity_ok', d['parity_ok'], d['quantile_error']['p50_rank_err_max'], d['quantile_error']['p99_rank_err_max'])"
grep -v amdgpu $O/err.txt | tail -3
