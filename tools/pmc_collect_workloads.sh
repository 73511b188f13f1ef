#!/bin/bash
# usage: tools/pmc_collect_workloads.sh <outdir-under-gpurun_out> [name ...]     (names: c2_conn c1 c5_zipf c3_levels; default the first three)
# Per sub-run of bench.py (the other BASELINE configurations, bench.py SUB_CONFIGS): a rocprofv3 kernel trace of the run restricted to its
# timed steps, and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) merged into
# <outdir>/pmc_traffic.json `workloads` by tools/pmc_workload.py (copy that file to profiles/pmc_traffic.json to have bench.py report it).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
NAMES=${@:-c2_conn c1 c5_zipf}
[ -f $O/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null  # (tools/profile_round.sh with the same outdir has just written the default workload's passes: keep them)
for name in $NAMES; do
  case $name in
    c2_conn) ARGS="--workload conn"; UNITS=$((1<<24)); UNIT=records ;;
    c1) ARGS="--hosts 1 --svcs 100 --events $((1<<26)) --nbuf 2"; UNITS=$((1<<26)); UNIT=events ;;
    c5_zipf) ARGS="--zipf-milli 1100 --hosts 50 --svcs 2000 --nbuf 2"; UNITS=$((1<<29)); UNIT=events ;;
    c3_levels) ARGS="--levels 1 --nbuf 2"; UNITS=$((1<<29)); UNIT=events ;;
    *) echo "unknown sub-run $name"; continue ;;
  esac
  K=10
  rm -rf /tmp/kt_$name /tmp/pf_$name /tmp/pw_$name
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o kt -- python $R/bench.py --sub $name $ARGS --no-quantile-check --steps $K --warmup 3 > $O/${name}_line_profiled.json 2> $O/${name}_kt.err
  for f in $(find /tmp/kt_$name -name "*.db"); do python $R/tools/rocprof_summary.py $f $O/${name}_kernel_stats.txt --timed $((K+1)) --anchor k_window_finish; done  # a step ends with the window close's k_window_finish: everything after the (K+1)-th-from-last one = the K timed steps
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$name -o p --output-format csv -- python $R/bench.py --sub $name $ARGS --no-quantile-check --steps 4 --warmup 2 > $O/${name}_pmc_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$name -o p --output-format csv -- python $R/bench.py --sub $name $ARGS --no-quantile-check --steps 4 --warmup 2 > $O/${name}_pmc_write.log 2>&1
  python $R/tools/pmc_workload.py /tmp/pf_$name /tmp/pw_$name $name 4 $O/pmc_traffic.json $UNITS $UNIT > $O/${name}_pmc_traffic_entry.json 2> $O/${name}_pmc_workload.err
  (echo "## $name: FETCH_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pf_$name gys::; echo "## $name: WRITE_SIZE pass"; python $R/tools/pmc_kernels.py /tmp/pw_$name gys::) >> $O/workloads_pmc_summary.txt 2>&1
  echo "== $name"; head -c 600 $O/${name}_line_profiled.json; echo; head -8 $O/${name}_kernel_stats.txt | cut -c1-160
done
