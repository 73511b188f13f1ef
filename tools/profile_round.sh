#!/bin/bash
# usage: tools/profile_round.sh <tag>   -- full measurement set for profiles/: default bench line, rocprofv3 kernel-trace stats of the
# same command, and the FETCH_SIZE / WRITE_SIZE PMC passes (each in its own run).  Everything lands in gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-quantile-check --no-host-fed --configs none --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/kt.err
for f in $(find /tmp/kt -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/kernel_stats.txt --timed 20; python $R/tools/rocprof_summary.py $f $O/kernel_stats_whole_run.txt; done
rm -rf /tmp/pf /tmp/pw
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --no-quantile-check --no-host-fed --configs none --steps 3 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --no-quantile-check --no-host-fed --configs none --steps 3 > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $((1<<29)) 10000000 3 $O/pmc_traffic.json
(echo "## FETCH_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pf gys::; echo "## WRITE_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pw gys::) > $O/pmc_summary.txt
head -c 1500 $O/bench_line.json; echo; head -12 $O/kernel_stats.txt | cut -c1-170
