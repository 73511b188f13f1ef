#!/bin/bash
# GPU call aa: kernel parameters read from the kernel-argument segment at their uses (merge kernel, event kernel) against the old form, two rounds; parity on the new default
cd /root/repo; O=gpurun_out/r6aa; mkdir -p $O
for r in 1 2; do tools/ab_libs.sh bench $O/r$r --steps 20 --warmup 5 > $O/ab$r.txt 2>&1; cat $O/ab$r.txt; done
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
