#!/bin/bash
# GPU call am: hipcc scheduler strategies on the whole library (max-ilp, max-memory-clause, AMDGPU trackers, relaxed occupancy) against the default build,
# quarter-size default workload (same values per key and window), two rounds on one box
cd /root/repo; O=gpurun_out/r6am; mkdir -p $O
for r in 1 2; do
 tools/ab_libs.sh bench $O/q_$r --hosts 2500 --events 134217728 --steps 10 --warmup 3 --configs none 2>&1
done | tee $O/ab.txt
