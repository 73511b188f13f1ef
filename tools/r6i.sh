#!/bin/bash
# round 6, call i: events through LDS with the next group's requests issued one group ahead (single buffer, no registers) against the strided loads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6i; mkdir -p $O
timeout 1200 python -m pytest tests/test_bench_launch.py tests/test_gpu_resp.py tests/test_gpu_round5.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
