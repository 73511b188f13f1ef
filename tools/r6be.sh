#!/bin/bash
# GPU call be: the records a merge folds into read ahead with its clusters and parked in LDS (default) against the read-modify-write at the merge's end (nopf); default workload at full size, three rounds; then the merge-related parity tests
cd /root/repo; O=gpurun_out/r6be; mkdir -p $O
for r in 1 2 3; do
 tools/ab_libs.sh bench $O/d_$r --steps 20 --warmup 5 --configs none 2>&1
done | tee $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "c3 or c1 or c5 or resp or levels or fold or hist" 2>&1 | tail -4 | tee $O/tests.txt
