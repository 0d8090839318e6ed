#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_nullable.py tests/test_gpu_misc.py tests/test_gpu_facade.py -m gpu -x -q > $O/tests_23.log 2>&1 || { tail -30 $O/tests_23.log; exit 1; }
tail -3 $O/tests_23.log
rm -f $O/rank_23.txt
for L in hip E4 E8; do
  echo "== lib $L" >> $O/rank_23.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/nullable_bench.py --bw 12 >> $O/rank_23.txt 2>&1 || exit 1
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/diag/rank_ab.py >> $O/rank_23.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/rank_23.txt
