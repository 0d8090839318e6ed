#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fle.py tests/test_gpu_fullsize.py tests/test_gpu_misc.py -m gpu -x -q > $O/tests_29.log 2>&1 || { tail -40 $O/tests_29.log; exit 1; }
tail -3 $O/tests_29.log
rm -f $O/ab_29.txt
for L in O A B hip O hip; do
  echo "== lib $L" >> $O/ab_29.txt
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 32,8 --what scan --sel 0.1 >> $O/ab_29.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/ab_29.txt
