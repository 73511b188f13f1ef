#!/bin/bash
# round 4, GPU call 5: A/B of the all-service histogram cells (per lane slot vs per wave), the Zipf shape with the wider prediction margin,
# the new queue-flusher test
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q -s 2>&1 | tail -8) > $O/pytest.log 2>&1; cat $O/pytest.log
tools/ab_libs.sh bench $O/ab_quarter --hosts 2500 --events 134217728 --steps 12 --warmup 3 --configs none > $O/ab_quarter.txt 2>&1
tools/ab_libs.sh bench $O/ab_full --steps 20 --warmup 5 --configs none > $O/ab_full.txt 2>&1
tools/ab_libs.sh bench $O/ab_full2 --steps 20 --warmup 5 --configs none > $O/ab_full2.txt 2>&1
cat $O/ab_quarter.txt $O/ab_full.txt $O/ab_full2.txt
rm -f gyeeta_amd/lib/libgysketch_ghwave.so
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %.2f G/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.02})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
timeout 300 python bench.py --sub c5_zipf --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 20 --warmup 5 --nbuf 2 > $O/c5.json 2> $O/c5.err; line $O/c5.json c5_zipf
timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 10 --warmup 3 --nbuf 2 > $O/c5_4000.json 2> $O/c5_4000.err; line $O/c5_4000.json c5_25x4000
(time timeout 300 python bench.py --sub c1 --hosts 1 --svcs 100 --events 67108864 --steps 20 --warmup 5 --nbuf 2 > $O/c1.json 2> $O/c1.err) 2>&1 | grep real; line $O/c1.json c1
