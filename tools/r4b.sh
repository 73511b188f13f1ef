#!/bin/bash
# round 4, GPU call 2: the filtered-query tests, which LDS access of k_resp_host owns its bank conflicts (one counter pass per switch-off
# build: GYS_RESP_DBG library + GYS_DBG launch switches), and the A/B of the merge's start stagger
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
(time timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_json.py -m gpu -x -q -s 2>&1 | tail -12) > $O/pytest.log 2>&1
cat $O/pytest.log
# ---- LDS bank conflicts by access: quarter size (2 500 hosts, 2^27 events per window: the per-key rates of the default line)
cd /tmp; export TMPDIR=/tmp
BARGS="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --hosts 2500 --events 134217728 --steps 4 --warmup 2"
for d in 0 16 8 24 2 1 4; do
  rm -rf /tmp/lds_$d
  GYS_LIB=$R/gyeeta_amd/lib/libgysketch_dbg.so GYS_DBG=$d timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/lds_$d -o p --output-format csv -- python $R/bench.py $BARGS > $O/lds_$d.log 2>&1
  echo "## GYS_DBG=$d (1 no flush, 2 no image, 4 no HLL, 8 no all-service histogram, 16 no key counts / rank atomics)" >> $O/lds_conflicts_by_access.txt
  python $R/tools/pmc_kernels.py /tmp/lds_$d "k_resp_host<16, false, false" >> $O/lds_conflicts_by_access.txt 2>&1
  tail -1 $O/lds_$d.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench (under the profiler): %.3f ms/step' % d['ms_per_step'], {k: round(v['ms'],3) for k,v in d['roofline']['kernels'].items() if v['ms']>0.05})
except Exception as e: print('   no line', e)" >> $O/lds_conflicts_by_access.txt
done
cat $O/lds_conflicts_by_access.txt
# ---- merge start stagger
cd $R
rm -f gyeeta_amd/lib/libgysketch_dbg.so
tools/ab_libs.sh bench $O/ab_quarter --hosts 2500 --events 134217728 --steps 12 --warmup 3 --configs none > $O/ab_quarter.txt 2>&1
tools/ab_libs.sh bench $O/ab_full --steps 20 --warmup 5 --configs none > $O/ab_full.txt 2>&1
cat $O/ab_quarter.txt $O/ab_full.txt
