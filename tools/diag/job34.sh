#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_misc.py tests/test_gpu_q6.py -m gpu -x -q > $O/tests_34.log 2>&1 || { tail -40 $O/tests_34.log; exit 1; }
tail -3 $O/tests_34.log
timeout -k 10 300 python tools/q6_bench.py > $O/q6_34.txt 2>&1
timeout -k 10 300 python tools/q6_bench.py >> $O/q6_34.txt 2>&1
grep -v amdgpu.ids $O/q6_34.txt
