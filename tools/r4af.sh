#!/bin/bash
# round 4, GPU call 31: k_fold's per-key meta broadcast, row reductions and row maximum on v_readlane / DPP instead of ds_bpermute (the
# fold of every key at a window close with gys_config.enable_levels = 1): default library against r4ae's kernels (libgysketch_r4ae) on
# `--levels 1`, `--levels 2` and the default line, twice each; then the GPU tests that fold (levels, resp, json, round2 / 3)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4af; mkdir -p $O; cd $R
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
for lib in libgysketch libgysketch_r4ae libgysketch libgysketch_r4ae; do
	for cfg in "levels1 --levels 1" "levels2 --levels 2" "c3"; do
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
(time timeout 900 python -m pytest tests -m gpu -x -q -k "levels or resp or json or round2 or round3 or day or scan" 2>&1 | tail -6) 2>&1 | grep -v amdgpu | tee $O/pytest.log
