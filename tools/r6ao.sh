#!/bin/bash
# GPU call ao: k_huge_count with 16-byte pieces requested ahead (GYS_HB_VEC 2 / 4 / 8, compiled for 4 or 8 waves per SIMD) against the one-word loop, C1 and C5 shapes, two rounds
cd /root/repo; O=gpurun_out/r6ao; mkdir -p $O
for r in 1 2; do
 tools/ab_libs.sh bench $O/c1_$r --hosts 1 --svcs 100 --events 67108864 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c1 /"
 tools/ab_libs.sh bench $O/c5_$r --zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2 --steps 10 --warmup 3 --configs none 2>&1 | sed "s/^/c5 /"
done | tee $O/ab.txt
