#!/bin/bash
# round 6, call a: the default line as the driver runs it (wall time!), then the headline-configuration parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a_bench_stdout.txt 2> gpurun_out/r6a_bench_stderr.txt ) 2> gpurun_out/r6a_bench_time.txt
echo "bench rc=$?"; tail -c 4200 gpurun_out/r6a_bench_stdout.txt; cat gpurun_out/r6a_bench_time.txt
wc -c gpurun_out/r6a_bench_stdout.txt
cp gpurun_out/bench_detail.json gpurun_out/r6a_bench_detail.json
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -x -q --durations=12 2>&1 | tail -30 | tee gpurun_out/r6a_tests.txt
