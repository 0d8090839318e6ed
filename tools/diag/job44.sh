#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_44.log 2>&1 || { tail -40 $O/tests_44.log; exit 1; }
tail -3 $O/tests_44.log
for i in 1 2; do timeout -k 10 300 python tools/configs_bench.py 2>&1 | grep "configs\[3\]" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['us_med'], round(d['GBps_med']/80,1))"; done
