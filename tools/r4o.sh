#!/bin/bash
# round 4, GPU call 15: the submission queues of the host-pointer connection / listener-state calls: the 16-thread parity test, the shim
# tests, then tools/cpp/bench_hostfed (16 L2 threads, pageable buffers) twice
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
(time timeout 600 python -m pytest tests -m gpu -x -q -k "16_threads" 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
timeout 300 python - <<'PY' 2>&1 | tee $O/hostfed.txt
import json, bench
for i in range(2):
    print(json.dumps(bench.host_fed_l2_threads()))
PY
