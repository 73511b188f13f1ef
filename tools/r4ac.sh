#!/bin/bash
# round 4, GPU call 28 (call 27 + a thread's eight buffered words combined per bucket before the LDS adds, four bins per lane in the scan): k_huge_merge with the conflict-free bin scan, eight buffered-word loads in flight and one-word cluster thresholds
# (default) against r4y's (libgysketch_hugeold) on C5 and C1, twice each; the phase ticks of the new kernel; the large-key parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ac; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --nbuf 2 --steps 10 --warmup 3"
for lib in libgysketch libgysketch_hugeold; do
	for cfg in "c5 --zipf-milli 1100 --hosts 50 --svcs 2000" "c1 --hosts 1 --svcs 100 --events 67108864"; do
		set -- $cfg; name=$1; shift
		GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 300 python bench.py "$@" $LEAN > $O/${name}_$lib.json 2> $O/${name}_$lib.err
		python - $O/${name}_$lib.json $name:$lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f G ev/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("parity_ok"), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
	done
done 2>&1 | tee $O/summary.txt
GYS_LIB=$R/gyeeta_amd/lib/libgysketch_hugetime.so timeout 300 python bench.py --zipf-milli 1100 --hosts 50 --svcs 2000 $LEAN > $O/c5_time.json 2> $O/c5_time.err
grep GYS_HUGE_TIMING $O/c5_time.err | tail -1 | tee -a $O/summary.txt
(time timeout 900 python -m pytest tests -m gpu -x -q -k "huge or zipf or c5 or c1 or configs or single_host or replay or large or resp" 2>&1 | tail -5) 2>&1 | grep -v amdgpu | tee $O/pytest.log
