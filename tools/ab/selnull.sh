#!/bin/bash
# dev tool: ips_dict_select_nullable -- parity, then timing by selectivity, one-pass route against
# the step-by-step one (IPS_SELECT_NULLABLE_STEPS=1)
mkdir -p gpurun_out/ab
timeout -k 10 900 python -m pytest tests/test_gpu_nullable.py tests/test_gpu_misc.py -m gpu -x -q -k "select_nullable or optional or nullable" > gpurun_out/ab/selnull_tests.log 2>&1 || { tail -40 gpurun_out/ab/selnull_tests.log; exit 1; }
tail -1 gpurun_out/ab/selnull_tests.log
echo "== one pass"; timeout -k 10 300 python tools/ab/selnull.py 2>&1 | grep -v amdgpu.ids
echo "== steps"; IPS_SELECT_NULLABLE_STEPS=1 timeout -k 10 300 python tools/ab/selnull.py 2>&1 | grep -v amdgpu.ids
