#!/bin/bash
# GPU call av: k_huge_count with the next step's pieces requested before this step's values are taken (p1: one piece per step, p2: two) against the default (two pieces, not pipelined); C1 and C5, two rounds
cd /root/repo; O=gpurun_out/r6av; mkdir -p $O
for r in 1 2; do
 tools/ab_libs.sh bench $O/c1_$r --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c1 /"
 tools/ab_libs.sh bench $O/c5_$r --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c5 /"
done | tee $O/ab.txt
