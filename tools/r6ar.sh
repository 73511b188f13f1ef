#!/bin/bash
# GPU call ar: kernel trace of the C1 and C5 shapes with the new k_huge_count (where the large-key path's time goes now)
R=/root/repo; O=$R/gpurun_out/r6ar; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --detail-out none --nbuf 2 --steps 10 --warmup 3"
rm -rf /tmp/k1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o kt -- python $R/bench.py $LEAN --hosts 1 --svcs 100 --events 67108864 > $O/c1.line 2> $O/c1.err
for f in $(find /tmp/k1 -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/c1_kernel_stats.txt --timed 10; done
rm -rf /tmp/k5; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k5 -o kt -- python $R/bench.py $LEAN --zipf-milli 1100 --hosts 50 --svcs 2000 > $O/c5.line 2> $O/c5.err
for f in $(find /tmp/k5 -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/c5_kernel_stats.txt --timed 10; done
head -12 $O/c1_kernel_stats.txt | cut -c1-170; head -14 $O/c5_kernel_stats.txt | cut -c1-170
