#!/bin/bash
# dev tool: alternating A/B of the nullable leaf on one box
mkdir -p gpurun_out/ab
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_nullable.py -m gpu -x -q > gpurun_out/ab/leaf_tests.log 2>&1 || { tail -30 gpurun_out/ab/leaf_tests.log; exit 1; }
tail -2 gpurun_out/ab/leaf_tests.log
echo "== fused"; timeout -k 10 300 python tools/ab/leaf_ab.py
echo "== separate"; IPS_NO_FUSED_LEAF=1 timeout -k 10 300 python tools/ab/leaf_ab.py
