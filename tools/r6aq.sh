#!/bin/bash
# GPU call aq: k_huge_count with the LDS reads of a thread's values staged (default here: 8 values; st1: 4) against the unstaged form (nostage) and round 5's loop (hbv0)
# then the C1 / C5 parity tests on the default library
cd /root/repo; O=gpurun_out/r6aq; mkdir -p $O
for r in 1 2; do
 tools/ab_libs.sh bench $O/c1_$r --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c1 /"
 tools/ab_libs.sh bench $O/c5_$r --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c5 /"
done | tee $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "c1 or c5 or huge or zipf or spill or heavy" 2>&1 | tail -4 | tee $O/tests.txt
