#!/bin/bash
# dev tool: a longer randomized soak (eight seeds of every soak; progress lines for the watchdog)
mkdir -p gpurun_out/soak
rc=0
for s in 31 32 33 34 35 36 37 38; do
  for t in soak soak_plain soak_nullable soak_dict soak_chunks soak_chain; do
    IPS_SOAK_SEED=$s IPS_SOAK_ITERS=400 timeout -k 10 600 python tools/$t.py > gpurun_out/soak/long_${t}_$s.log 2>&1 || rc=1
    echo "$t seed $s: $(tail -1 gpurun_out/soak/long_${t}_$s.log)"
  done
done
exit $rc
