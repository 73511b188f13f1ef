#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3w}; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_conn_lstate.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_wire.py -x -q 2>&1 | tail -5) | tee $O/pytest.log
for lib in gyeeta_amd/lib/libgysketch.so $(ls gyeeta_amd/lib/libgysketch_cs*.so 2>/dev/null); do
	tag=$(basename $lib .so)
	for st in messages mixed; do
	GYS_LIB=$R/$lib timeout 120 python bench.py --workload conn --conn-stream $st --no-cpu-baseline --no-host-fed --steps 10 --warmup 3 > $O/${tag}_$st.json 2> $O/${tag}_$st.err
	python - $O/${tag}_$st.json ${tag}_$st <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.2f G rec/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
	done
done
