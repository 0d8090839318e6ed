#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_39.txt
for L in hip W4 hip W4; do
  echo "== lib $L" >> $O/ab_39.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 32 --what scan --sel 0.1,0.01,0.3 >> $O/ab_39.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/ab_39.txt
