#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_30.log 2>&1 || { tail -40 $O/tests_30.log; exit 1; }
tail -3 $O/tests_30.log
timeout -k 10 300 python tools/aux_bench.py > $O/aux_30.txt 2>&1
grep -v amdgpu.ids $O/aux_30.txt
