#!/bin/bash
# round 4, GPU call 16: all GPU tests on the tree with the record submission queues (the two 16-thread tests five times over), smoke(),
# and the default bench line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests -m gpu -x -q -k "16_threads" 2>&1 | tail -1; done | tee $O/threads_x5.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
(time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err) 2>&1 | tail -3
head -c 600 $O/bench_line.json; echo
