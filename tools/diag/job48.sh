#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_48.txt
for L in hip M16 hip M16; do
  echo "== lib $L" >> $O/ab_48.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 10,12,16 --what scan --sel 0.1,0.01,0.5,1.0 >> $O/ab_48.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/ab_48.txt
