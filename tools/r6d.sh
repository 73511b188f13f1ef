#!/bin/bash
# round 6, call d: next-group line prefetch A/B; where the event kernel's wave time goes (s_memtime ticks per phase, -DGYS_RESP_TIMING build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resp.py tests/test_gpu_round5.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
grep -h GYS_RESP_TIMING $O/ab/libgysketch_timing.err | tail -2 | tee $O/timing.txt
