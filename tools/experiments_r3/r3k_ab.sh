#!/bin/bash
# round-3 A/B of the restructured event kernel: whole GPU suite on the default library, then the default line / the quarter-size line
# for every library under gyeeta_amd/lib (libgysketch.so first) and the 12-event tile form of the default one
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3k}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %.2f G ev/s %.3f ms parity=%s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d.get("parity_ok")), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { tag=$1; lib=$2; shift 2; GYS_LIB=$R/$lib timeout 280 python bench.py --no-cpu-baseline --no-host-fed "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json $tag; }
for lib in gyeeta_amd/lib/libgysketch.so $(ls gyeeta_amd/lib/libgysketch_*.so 2>/dev/null); do
	tag=$(basename $lib .so)
	run ${tag}_default $lib --steps 20 --warmup 5
	run ${tag}_quarter $lib --hosts 2500 --events 134217728 --steps 10 --warmup 3 --no-quantile-check
done
GYS_TPT=12 run tpt12_default gyeeta_amd/lib/libgysketch.so --steps 20 --warmup 5 --no-quantile-check
run c5_50x2000 gyeeta_amd/lib/libgysketch.so --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2
run c5_25x4000 gyeeta_amd/lib/libgysketch.so --zipf-milli 1100 --hosts 25 --svcs 4000 --steps 8 --warmup 2
