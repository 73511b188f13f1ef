#!/bin/bash
# round 4, GPU call 33: last check of the committed tree (library rebuilt from it: build commit = HEAD's kernels, device code
# 2d13edb1b1765535): all GPU tests, smoke(), the default line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ah; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) 2>&1 | grep -v amdgpu | tee $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
(time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err) 2>&1 | tail -3 | tee $O/bench_time.txt
python - $O/bench_line.json <<'PY' | tee $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["build_commit"], d["device_code"], "%.2f G events/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), "parity_ok", d["parity_ok"], "frac", round(d["roofline"]["frac"], 3),
      "traffic of these kernels:", all((v.get("traffic") or {}).get("same_kernels_as_this_run") in (True, None) for v in d["roofline"]["kernels"].values()))
for n, c in d["configs"].items():
    print(" ", n, "%.2f G %s %.3f ms frac %.3f" % (c["value"] / 1e9, c["unit"], c["ms_per_step"], c["roofline"]["frac"]))
PY
