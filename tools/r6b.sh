#!/bin/bash
# round 6, call b: the event-phase diet -- parity (resp tests + config tests) on the new default library, then A/B lines per switch,
# then the default line as the driver runs it (wall time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resp.py tests/test_gpu_round5.py tests/test_gpu_configs.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt ) 2> $O/bench_time.txt
tail -c 3000 $O/bench_stdout.txt; cat $O/bench_time.txt
cp gpurun_out/bench_detail.json $O/bench_detail.json
