#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_45.log 2>&1 || { tail -40 $O/tests_45.log; exit 1; }
tail -3 $O/tests_45.log
rm -f $O/ab_45.txt
for L in prev hip prev hip; do
  echo "== lib $L" >> $O/ab_45.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 32,16,8 --what scan,pred,decode --sel 0.1 >> $O/ab_45.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/q6_bench.py 2>&1 | grep per-operand >> $O/ab_45.txt
done
grep -v amdgpu.ids $O/ab_45.txt
