#!/bin/bash
# A/B of whole TREES (constants shared with the oracle differ, so each tree carries its own library + oracle) inside ONE gpurun call.
#   here:        git worktree add _ab/<name> <branch>; (cd _ab/<name> && python -c 'import __graft_entry__ as g; g.build()')
#   on the box:  tools/ab_trees.sh OUTDIR tree1 tree2 ...      ("." = the main tree)
# per tree: GPU parity tests, the default bench line (lean), the per-rank loads of the N = 2 / 4 / 8 runs.
TOP=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$TOP/$1; shift
mkdir -p $OUT
for t in "$@"; do
	name=$(basename $(cd $TOP/$t && pwd)); [ "$t" = "." ] && name=main
	export GRAFT_REPO_ROOT=$(cd $TOP/$t && pwd)
	cd $GRAFT_REPO_ROOT
	echo "=== tree $name"
	(timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) | tee $OUT/$name.pytest.log
	timeout 200 python bench.py --no-cpu-baseline --no-host-fed --steps 10 --warmup 2 > $OUT/$name.bench.json 2> $OUT/$name.bench.err
	python - $OUT/$name.bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("default: %.2f G ev/s %.3f ms parity_ok=%s qerr=%s" % (d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok"), d.get("quantile_error")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print("bench failed:", e)
PY
	bash tools/regimes.sh 5000 2500 1250
done
