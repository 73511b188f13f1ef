#!/bin/bash
# round 6, call f: which unit does the event kernel wait for?  TA / TCP / TCC counter passes (names checked against `rocprofv3 -L` first)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_all.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -oE "\b(TA|TCP|TD|TCC|SQC|SPI|SQ)_[A-Za-z0-9_]+" $O/counters_all.txt | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
have() { for n in "$@"; do grep -qx "$n" $O/counter_names.txt && printf "%s " "$n"; done; }
S1=$(have TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum)
S2=$(have TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum)
S3=$(have TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum)
S4=$(have TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum)
S5=$(have TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_sum)
S6=$(have TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TD_ATOMIC_WAVEFRONT_sum TD_SPI_STALL_sum)
S7=$(have SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES)
S8=$(have SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL)
S9=$(have SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_TC_STALL)
echo "S1=$S1"; echo "S2=$S2"; echo "S3=$S3"; echo "S4=$S4"; echo "S5=$S5"; echo "S6=$S6"; echo "S7=$S7"; echo "S8=$S8"; echo "S9=$S9"
ARGS="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --steps 3 --warmup 2 --prime-windows 2"
sets=()
for s in "$S1" "$S2" "$S3" "$S4" "$S5" "$S6" "$S7" "$S8" "$S9"; do [ -n "$s" ] && sets+=("$s"); done
bash tools/pmc_collect.sh r6f "$ARGS" "${sets[@]}" 2>&1 | grep -E "^## pmc|k_resp_host<16, false, false|k_digest_bins<false, 8" | tee $O/summary_hot.txt
