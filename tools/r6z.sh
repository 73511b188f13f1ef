#!/bin/bash
# GPU call z: roll-up with eight counters per hot value (LDS same-address adds) -- parity and time
cd /root/repo; O=gpurun_out/r6z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for r in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --steps 6 --warmup 2 --detail-out $O/bench$r.json > $O/bench$r.line 2> $O/bench$r.err
python - $r <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6z/bench%s.json" % sys.argv[1]))
print(d["value"] / 1e9, d["ms_per_step"], {k: v for k, v in d.get("quantile_scan", {}).items() if "rollup" in k})
PY
done
