#!/bin/bash
# full GPU suite + the multi-device tests first (fast feedback in the log tail order)
cd /root/repo
mkdir -p gpurun_out
S=$SECONDS
timeout 900 python -m pytest tests/test_gpu_multi_device.py -q -x 2>&1 | tail -25 > gpurun_out/r02_multi.log
echo "multi wall $((SECONDS-S)) s" >> gpurun_out/r02_multi.log
cat gpurun_out/r02_multi.log
S=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multi_device.py 2>&1 | tail -40 > gpurun_out/r02_gpu_suite.log
echo "suite wall $((SECONDS-S)) s" >> gpurun_out/r02_gpu_suite.log
cat gpurun_out/r02_gpu_suite.log
