#!/bin/bash
O=gpurun_out/r2
mkdir -p $O
rm -f $O/ab_46.txt
for L in prev hip prev hip prev hip; do
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_$L.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$L', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min_of_5_after_region'])" >> $O/ab_46.txt
done
cat $O/ab_46.txt
