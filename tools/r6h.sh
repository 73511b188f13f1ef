#!/bin/bash
# round 6, call h: the wave's events through LDS (global_load_lds_dwordx4, every line requested once) against the strided 16 + 8-byte loads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resp.py tests/test_gpu_round5.py tests/test_gpu_configs.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
