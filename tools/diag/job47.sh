#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_47.log 2>&1 || { tail -40 $O/tests_47.log; exit 1; }
tail -3 $O/tests_47.log
timeout -k 10 300 python tools/kbench.py --bw 32 --what pred --sel 0.1 2>&1 | grep pred
timeout -k 10 300 python tools/configs_bench.py 2>&1 | grep "PLAIN" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['us_med'], round(d['GBps_med']/80,1))"
