#!/bin/bash
# round 6, call k: the round's own GPU tests (state decision, two-level global roll-up) + the roll-up tests of round 2; events through LDS with
# non-temporal requests (the event lines leave the L2 to the flush's partly written buffer lines); the default line's query-boundary figures
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q 2>&1 | tail -6 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --configs none --steps 10 --warmup 3 --detail-out $O/scan_detail.json > $O/scan_line.json 2> $O/scan.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6k/scan_detail.json"))
print("quantile_scan", d.get("quantile_scan"))
PY
