#!/bin/bash
# round-3 evidence on the final tree: default bench line (driver flags), rocprofv3 kernel trace restricted to the timed region,
# FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic.json, and two SQ / LDS counter sets for the shipped kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/profile_round.sh r3an
bash $R/tools/pmc_collect.sh r3an "--no-cpu-baseline --no-host-fed --no-quantile-check --steps 3 --warmup 2" \
	"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
	"SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > /dev/null
grep -E "pmc set|k_resp_host<16, false, false|k_digest_bins" $R/gpurun_out/r3an/pmc_summary.txt | cut -c1-400
