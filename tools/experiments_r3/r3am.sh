#!/bin/bash
# r3am: H2D copies of the response submissions on their own stream (default) against everything on the engine stream (GYS_RQ_ONE_STREAM=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r3am}; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_cpp_shim.py tests/test_gpu_conn_lstate.py tests/test_gpu_round2.py -x -q 2>&1 | tail -3) | tee $O/pytest.log
g++ -std=c++17 -O2 tools/cpp/bench_hostfed.cc -o /tmp/bench_hostfed -Lgyeeta_amd/lib -lgysketch -Wl,-rpath,$R/gyeeta_amd/lib -Wl,-rpath,/opt/rocm/lib -pthread
for v in copy_stream one_stream copy_stream one_stream; do
	if [ $v = one_stream ]; then export GYS_RQ_ONE_STREAM=1; else unset GYS_RQ_ONE_STREAM; fi
	/tmp/bench_hostfed 16 2 > $O/$v.json 2> $O/$v.err
	python - $O/$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s resp %.3f G ev/s (%.1f GB/s)  conn %.1f M rec/s  lstate %.1f M rec/s  calls/submission %.1f" % (sys.argv[2], d["resp_events"]["records_per_s"] / 1e9, d["resp_events"]["GBps"], d["tcp_conn"]["records_per_s"] / 1e6, d["listener_state"]["records_per_s"] / 1e6, d["counters"]["resp_calls_queued"] / max(1, d["counters"]["resp_submissions"])))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
