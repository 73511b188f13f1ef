#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r3z}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o kt -- python $R/bench.py --no-cpu-baseline --no-quantile-check --no-host-fed --zipf-milli 1100 --hosts 50 --svcs 2000 --steps 8 --warmup 2 > $O/c5_profiled.json 2> $O/kt.err
for f in $(find /tmp/kt5 -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/c5_kernel_stats.txt --timed 8; done
head -24 $O/c5_kernel_stats.txt | cut -c1-160
