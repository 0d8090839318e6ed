#!/bin/bash
# dev tool: the headline scan complete / phase B without its value stores / without phase B
# (build the variants first: make -C impala-avx2-parquet-scanner_amd/csrc OUT=../libabl4.so BUILD=build_abl4 EXTRA=-DIPS_ABLATE=4, same with 3)
for i in 1 2 3; do
for lib in libips_hip.so libabl4.so libabl3.so; do
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/$lib timeout -k 10 300 python tools/ab/ablate_headline.py 2>&1 | grep -v amdgpu
done
done
