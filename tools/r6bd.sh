#!/bin/bash
# GPU call bd: soak -- the whole GPU suite twice more on the final tree (another box than r6al), then three default lean lines (spread)
cd /root/repo; O=gpurun_out/r6bd; mkdir -p $O
for r in 1 2; do timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gpu_tests_$r.txt; cat $O/gpu_tests_$r.txt; done
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 20 --warmup 5 --detail-out none 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f G events/s %.3f ms resp_host %.3f ms'%(d['value']/1e9,d['ms_per_step'],d['roofline']['kernel_avg_ms']))"; done | tee $O/spread.txt
