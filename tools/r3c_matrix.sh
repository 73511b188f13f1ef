#!/bin/bash
# round-3 experiment matrix (one gpurun call): compile-time variants (libgysketch_<tag>.so from tools/ab_libs.sh build) x launch forms
# (GYS_RESP_DIRECT, GYS_TPT), each a lean default-size bench line; then the parity tests on the candidate combination
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r3c; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
run() { # tag lib env...
	tag=$1; lib=$2; shift 2
	env GYS_LIB=$R/gyeeta_amd/lib/$lib "$@" timeout 150 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --steps 20 --warmup 3 > $O/$tag.json 2> $O/$tag.err
	line $O/$tag.json $tag
}
run base            libgysketch.so
run base_tpt12      libgysketch.so GYS_TPT=12
run base_direct1024 libgysketch.so GYS_RESP_DIRECT=1024
run base_direct512  libgysketch.so GYS_RESP_DIRECT=512
run base_direct256  libgysketch.so GYS_RESP_DIRECT=256
run lut             libgysketch_lut.so
run pk              libgysketch_pk.so
run all_direct1024  libgysketch_all.so GYS_RESP_DIRECT=1024
run all_direct512   libgysketch_all.so GYS_RESP_DIRECT=512
run all             libgysketch_all.so
echo "== parity: all + direct1024"
(GYS_LIB=$R/gyeeta_amd/lib/libgysketch_all.so GYS_RESP_DIRECT=1024 timeout 300 python -m pytest tests/test_gpu_resp.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5)
echo "== parity: all + direct512, quantile check"
GYS_LIB=$R/gyeeta_amd/lib/libgysketch_all.so GYS_RESP_DIRECT=512 timeout 200 python bench.py --no-cpu-baseline --no-host-fed --steps 10 --warmup 3 > $O/all_direct512_q.json 2> $O/all_direct512_q.err
python -c "
import json; d=json.loads(open('$O/all_direct512_q.json').read().strip().splitlines()[-1]); print('parity_ok', d.get('parity_ok'), d.get('quantile_error'))"
echo "== round-3 tests on the default library"
(timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -5)
echo "== host-fed (submission queue) on the default library"
g++ -std=c++17 -O2 tools/cpp/bench_hostfed.cc -o /tmp/bhf -Lgyeeta_amd/lib -lgysketch -Wl,-rpath,$R/gyeeta_amd/lib -Wl,-rpath,/opt/rocm/lib -pthread && timeout 120 /tmp/bhf 16 2.0 | tee $O/hostfed.json
