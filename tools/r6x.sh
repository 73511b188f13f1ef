#!/bin/bash
# GPU call x: roll-up with the member identity read ahead -- parity, time, and the kernels' own durations (rocprofv3 kernel trace of the same bench command)
cd /root/repo; O=gpurun_out/r6x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py -m gpu -x -q -k "rollup" 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for r in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --steps 6 --warmup 2 --detail-out $O/bench$r.json > $O/bench$r.line 2> $O/bench$r.err
python - $r <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6x/bench%s.json" % sys.argv[1]))
print(d["value"] / 1e9, d["ms_per_step"], {k: v for k, v in d.get("quantile_scan", {}).items() if "rollup" in k})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o rollup -- python /root/repo/bench.py --no-cpu-baseline --no-host-fed --steps 4 --warmup 1 --detail-out /root/repo/$O/bench_prof.json > /root/repo/$O/prof.log 2>&1
cd /root/repo
f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
grep -i "rollup\|Name" $f | head -8
cp $f $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
