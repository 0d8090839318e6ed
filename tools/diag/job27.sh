#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_27.log 2>&1 || { tail -40 $O/tests_27.log; exit 1; }
tail -3 $O/tests_27.log
timeout -k 10 300 python tools/configs_bench.py > $O/configs_27.txt 2>&1
grep -v amdgpu.ids $O/configs_27.txt | head -60
