#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_fle.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do for L in hip G0 G3 G8; do echo "== $L"; IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python tools/kbench.py --bw 4,8 --what scan --sel 0.001,0.01,0.03,0.1 2>&1 | grep scan | cut -c1-60; done; done
