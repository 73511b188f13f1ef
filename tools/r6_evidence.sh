#!/bin/bash
# usage: tools/r6_evidence.sh <tag>   -- round-6 evidence on the tree as it is: HBM counter passes of the default workload and of the four sub-runs FIRST
# (so that the line printed afterwards cites the traffic of the very kernels it ran: device_code on both sides), the timed-region kernel trace, the SQ / LDS
# counter sets, smoke(), then the default line exactly as the driver runs it (stdout = the compact line; the detail file next to it)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --detail-out none"
rm -rf /tmp/pf /tmp/pw
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o p --output-format csv -- python $R/bench.py $LEAN --steps 3 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o p --output-format csv -- python $R/bench.py $LEAN --steps 3 > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $((1<<29)) 10000000 3 $O/pmc_traffic.json > /dev/null
(echo "## FETCH_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pf gys::; echo "## WRITE_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pw gys::) > $O/pmc_fetch_write_summary.txt
bash $R/tools/pmc_collect_workloads.sh $T c2_conn c1 c5_zipf c3_levels > $O/workloads.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py $LEAN --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/kt.err
for f in $(find /tmp/kt -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/kernel_stats.txt --timed 20; python $R/tools/rocprof_summary.py $f $O/kernel_stats_whole_run.txt; done
bash $R/tools/pmc_collect.sh $T "$LEAN --steps 3 --warmup 2" \
	"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
	"SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > /dev/null
(cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt)
(cd $R && time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err) > $O/bench_time.txt 2>&1
tail -c 3000 $O/bench_line.json; echo; head -8 $O/kernel_stats.txt | cut -c1-170; cat $O/bench_time.txt; grep -E "pmc set|k_resp_host<16, false, false, false|k_digest_bins<false" $O/pmc_summary.txt | cut -c1-520; tail -32 $O/workloads.txt | cut -c1-200
