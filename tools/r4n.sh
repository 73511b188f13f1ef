#!/bin/bash
# round 4, GPU call 14 (after the conn ingest at twelve waves with planned spans and the listener-state LDS roll-up): all GPU parity tests, then the evidence set (tools/r4_evidence.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
bash tools/r4_evidence.sh r4n
