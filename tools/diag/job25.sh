#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_25.txt
for i in 1 2; do
for L in prev hip; do
  echo "== lib $L" >> $O/ab_25.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/q6_bench.py >> $O/ab_25.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 12,8,4 --what pred >> $O/ab_25.txt 2>&1
done
done
grep -v amdgpu.ids $O/ab_25.txt
