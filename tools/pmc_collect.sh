#!/bin/bash
# usage: tools/pmc_collect.sh <outdir-under-gpurun_out> "<bench args>" "<counter set 1>" ["<counter set 2>" ...]
# one rocprofv3 --pmc pass per counter set (no trace domains besides --kernel-trace), per-kernel averages appended to pmc_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
BARGS=$1; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o p --output-format csv -- python $R/bench.py $BARGS > $O/pmc_$i.log 2>&1
  echo "## pmc set: $set   (bench.py $BARGS)" >> $O/pmc_summary.txt
  python $R/tools/pmc_kernels.py /tmp/pmc_$i gys:: >> $O/pmc_summary.txt 2>&1
done
cat $O/pmc_summary.txt
