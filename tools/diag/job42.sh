#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_42.txt
for L in hip N P hip N P; do
  echo "== lib $L" >> $O/ab_42.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 32,8 --what decode >> $O/ab_42.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/ab_42.txt
