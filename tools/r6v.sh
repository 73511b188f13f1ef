#!/bin/bash
# GPU call v: the early-issue event loads (GYS_EV_PIPE 1 / 2) against the default kernels, two rounds each, plus the resp tests on the variants
cd /root/repo; O=gpurun_out/r6v; mkdir -p $O
for r in 1 2; do tools/ab_libs.sh bench $O/r$r --steps 20 --warmup 5 > $O/ab$r.txt 2>&1; cat $O/ab$r.txt; done
for t in pipe1 pipe2; do GYS_LIB=/root/repo/gyeeta_amd/lib/libgysketch_$t.so timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c3 or c1" 2>&1 | tail -3 > $O/tests_$t.txt; cat $O/tests_$t.txt; done
