#!/bin/bash
# GPU call au: SQ / LDS counters of the large-key path on the C1 shape (what bounds k_huge_count after r6ao) + FETCH_SIZE
cd /root/repo
bash tools/pmc_collect.sh r6au "--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --detail-out none --nbuf 2 --steps 3 --warmup 2 --hosts 1 --svcs 100 --events 67108864" \
	"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
	"SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
	"FETCH_SIZE" "GRBM_GUI_ACTIVE" 2>&1 | grep -E "pmc set|k_huge|k_resp_host<12, true, false" | cut -c1-420
