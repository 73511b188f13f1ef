#!/bin/bash
# round 4, GPU call 30 (final kernels of the round: DPP wave scans, large-key merge, conn ingest at sixteen waves): all GPU parity tests, then the evidence set (tools/r4_evidence.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ae; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
bash tools/r4_evidence.sh r4ae
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2 | tee $O/smoke.log
