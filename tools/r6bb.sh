#!/bin/bash
# GPU call bb: the whole GPU suite on the final tree, then the evidence set (counters, kernel trace, smoke, the driver-style line)
cd /root/repo; O=gpurun_out/r6bb; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 2400 bash tools/r6_evidence.sh r6bb 2>&1 | tail -60
