#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_38.log 2>&1 || { tail -40 $O/tests_38.log; exit 1; }
tail -3 $O/tests_38.log
rm -f $O/ab_38.txt
for L in S0 hip S0 hip; do
  echo "== lib $L" >> $O/ab_38.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 4,8,12,16 --what scan --sel 0.1,0.01,0.5,1.0 >> $O/ab_38.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/configs_bench.py 2>&1 | grep "configs\[3\]" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['us_med'], round(d['GBps_med']/80,1))" >> $O/ab_38.txt
done
grep -v amdgpu.ids $O/ab_38.txt
