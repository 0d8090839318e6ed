#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_nullable.py tests/test_gpu_misc.py tests/test_gpu_facade.py -m gpu -x -q > $O/tests_20.log 2>&1 || { tail -30 $O/tests_20.log; exit 1; }
tail -3 $O/tests_20.log
timeout -k 10 300 python tools/nullable_bench.py --bw 12 > $O/rank_20.txt 2>&1 || exit 1
timeout -k 10 300 python tools/diag/rank_ab.py >> $O/rank_20.txt 2>&1
cat $O/rank_20.txt
