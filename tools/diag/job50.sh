#!/bin/bash
set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_50.log 2>&1 || { tail -40 $O/tests_50.log; exit 1; }
tail -3 $O/tests_50.log
for L in DP0 hip DP0 hip; do echo "== $L"; IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 400 python tools/diag/dec_ab.py 2>&1 | grep -v amdgpu; done
