#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_24.log 2>&1 || { tail -30 $O/tests_24.log; exit 1; }
tail -3 $O/tests_24.log
timeout -k 10 300 python tools/nullable_bench.py --bw 12,8 --nulls 0.1,0.5 > $O/rank_24.txt 2>&1 || exit 1
timeout -k 10 300 python tools/q6_bench.py >> $O/rank_24.txt 2>&1
grep -v amdgpu.ids $O/rank_24.txt
