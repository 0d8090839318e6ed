#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_35.log 2>&1 || { tail -40 $O/tests_35.log; exit 1; }
tail -3 $O/tests_35.log
timeout -k 10 300 python tools/q6_bench.py 2>&1 | grep Q6
