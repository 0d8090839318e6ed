#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_33.log 2>&1 || { tail -40 $O/tests_33.log; exit 1; }
tail -3 $O/tests_33.log
rm -f $O/ab_33.txt
for L in T1 hip T1 hip; do
  echo "== lib $L" >> $O/ab_33.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 4,6,8,12,16,20 --what pred >> $O/ab_33.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/q6_bench.py 2>&1 | grep per-operand >> $O/ab_33.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/nullable_bench.py --bw 12 2>&1 | grep "nullable leaf" >> $O/ab_33.txt
done
grep -v amdgpu.ids $O/ab_33.txt
