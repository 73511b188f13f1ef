#!/bin/bash
# round 5, last evidence call (after the IPv6 flow-hash change: the default instance's ISA is unchanged, the library's device code hash is not):
# the HBM counter passes of the default workload and of the four sub-runs, the timed-region kernel trace, then the line that cites them
# (tools/r5_evidence.sh without the SQ counter sets and without the CPU legs: the GPU minutes left for the round)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5u; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none"
rm -rf /tmp/pf /tmp/pw
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o p --output-format csv -- python $R/bench.py $LEAN --steps 3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o p --output-format csv -- python $R/bench.py $LEAN --steps 3 > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $((1<<29)) 10000000 3 $O/pmc_traffic.json > /dev/null
(echo "## FETCH_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pf gys::; echo "## WRITE_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pw gys::) > $O/pmc_fetch_write_summary.txt
bash $R/tools/pmc_collect_workloads.sh r5u c2_conn c1 c5_zipf c3_levels > $O/workloads.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py $LEAN --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/kt.err
for f in $(find /tmp/kt -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/kernel_stats.txt --timed 20; python $R/tools/rocprof_summary.py $f $O/kernel_stats_whole_run.txt; done
(cd $R && time timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err) > $O/bench_time.txt 2>&1
head -c 500 $O/bench_line.json; echo; head -6 $O/kernel_stats.txt | cut -c1-170; cat $O/bench_time.txt; grep -A5 "^== " $O/workloads.txt | cut -c1-170
