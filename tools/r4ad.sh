#!/bin/bash
# round 4, GPU call 29: wave scans / reductions of the hot kernels on DPP (k_resp_host's per-tile key-count scan, k_digest_bins' three scans
# and its min / max, the large-key path) + the large-key merge work of calls 26 - 28: default library against r4y's kernels
# (libgysketch_hugeold) on C3 (default line, lean), C5 and C1, twice each; then ALL GPU parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ad; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
for lib in libgysketch libgysketch_hugeold libgysketch libgysketch_hugeold; do
	for cfg in "c3" "c5 --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2" "c1 --hosts 1 --svcs 100 --events 67108864 --nbuf 2"; do
		set -- $cfg; name=$1; shift
		GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 300 python bench.py "$@" $LEAN > $O/${name}_$lib.json 2> $O/${name}_$lib.err
		python - $O/${name}_$lib.json $name:$lib <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f G ev/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
	done
done 2>&1 | tee $O/summary.txt
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) 2>&1 | grep -v amdgpu | tee $O/pytest.log
