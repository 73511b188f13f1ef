#!/bin/bash
# round 4, GPU call 24 (final kernels of the round: conn ingest at sixteen waves without global reads in its walk): all GPU parity tests, then the evidence set (tools/r4_evidence.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
bash tools/r4_evidence.sh r4y
