#!/bin/bash
# round 6, call g: the whole GPU suite on the current tree; then TA / TCP / TD busy and stall counters of the event kernel, two per pass
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 | tee $O/tests.txt
ARGS="--no-cpu-baseline --no-quantile-check --no-host-fed --configs none --steps 3 --warmup 2 --prime-windows 2"
bash tools/pmc_collect.sh r6g "$ARGS" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
   "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" 2>&1 | grep -E "^## pmc|k_resp_host<16, false, false|k_digest_bins<false, 8" | tee $O/summary_hot.txt
grep -m2 -i "exceeds\|error code" $O/pmc_*.log | cut -c1-200
