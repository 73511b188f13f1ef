#!/bin/bash
# GPU call ak: k_huge_merge with its parameters from the kernel-argument segment against the old form, on the C5 and C1 shapes (two rounds)
cd /root/repo; O=gpurun_out/r6ak; mkdir -p $O
for r in 1 2; do
 tools/ab_libs.sh bench $O/c5_$r --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --steps 10 --warmup 3 2>&1 | sed "s/^/c5 /"
 tools/ab_libs.sh bench $O/c1_$r --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --steps 10 --warmup 3 2>&1 | sed "s/^/c1 /"
done | tee $O/ab.txt
