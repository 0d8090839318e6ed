#!/bin/bash
# A/B of the rank kernels' tile size (IPS_RANK_ROUNDS = 8 / 4 / 2): nullable leaf, expand, compress
set -x
O=gpurun_out/r2
mkdir -p $O
for L in hip R4 R2; do
  echo "== lib $L" >> $O/rank_19.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/nullable_bench.py --bw 12 >> $O/rank_19.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/diag/rank_ab.py >> $O/rank_19.txt 2>&1 || exit 1
done
cat $O/rank_19.txt
