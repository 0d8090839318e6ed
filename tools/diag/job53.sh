#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_misc.py tests/test_gpu_facade.py -m gpu -x -q > $O/tests_53.log 2>&1 || { tail -40 $O/tests_53.log; exit 1; }
tail -3 $O/tests_53.log
timeout -k 10 300 python tools/enc_bench.py 2>&1 | grep -v amdgpu | tail -12
