#!/bin/bash
# GPU call az: kernel trace of the C5 shape with class 2 through the streamed value-bin instance
R=/root/repo; O=$R/gpurun_out/r6az; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --detail-out none --nbuf 2 --steps 10 --warmup 3"
rm -rf /tmp/k5; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k5 -o kt -- python $R/bench.py $LEAN --zipf-milli 1100 --hosts 50 --svcs 2000 > $O/c5.line 2> $O/c5.err
for f in $(find /tmp/k5 -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/c5_kernel_stats.txt --timed 10 > /dev/null 2>&1; done
head -16 $O/c5_kernel_stats.txt | cut -c1-170
