#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ablate_28.txt
for L in hip A3 A1 A2 hip; do
  echo "== lib $L" >> $O/ablate_28.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 32 --what scan --sel 0.1,0.01 >> $O/ablate_28.txt 2>&1 || exit 1
done
IPS_NO_EARLY_PRUNE=1 timeout -k 10 300 python tools/kbench.py --bw 32 --what pred >> $O/ablate_28.txt 2>&1
grep -v amdgpu.ids $O/ablate_28.txt
