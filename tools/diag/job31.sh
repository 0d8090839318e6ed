#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_31.txt
for L in prev hip prev hip; do
  echo "== lib $L" >> $O/ab_31.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/aux_bench.py 2>&1 | grep "compact\|assemble" >> $O/ab_31.txt
done
cat $O/ab_31.txt
