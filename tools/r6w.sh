#!/bin/bash
# GPU call w: the roll-up by value bin -- parity tests, then the time of the global roll-up at 10^7 services (bench.py's quantile_scan block)
cd /root/repo; O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round6.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --detail-out $O/bench.json > $O/bench.line 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6w/bench.json"))
print(d["value"] / 1e9, d["ms_per_step"], d.get("quantile_scan"))
PY
