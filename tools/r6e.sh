#!/bin/bash
# round 6, call e: prefetch issued AFTER the group's loads have arrived; HLL floor from a quarter of the register file per tile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resp.py tests/test_gpu_round5.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
