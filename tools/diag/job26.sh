#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_26.log 2>&1 || { tail -30 $O/tests_26.log; exit 1; }
tail -3 $O/tests_26.log
timeout -k 10 300 python tools/nullable_bench.py --bw 12,8 --nulls 0.1,0.5 > $O/rank_26.txt 2>&1 || exit 1
grep -v amdgpu.ids $O/rank_26.txt
