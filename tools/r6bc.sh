#!/bin/bash
# GPU call bc: the new merge-size-class test (both front ends, both buffer sizes) + the response-path file it lives in
cd /root/repo; O=gpurun_out/r6bc; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resp.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
