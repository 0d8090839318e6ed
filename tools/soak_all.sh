#!/bin/bash
# dev tool: every randomized differential soak, three seeds each (about two minutes on one GPU)
mkdir -p gpurun_out/soak
rc=0
for s in 11 12 13; do
  for t in soak soak_plain soak_nullable soak_dict soak_chunks soak_chain; do
    IPS_SOAK_SEED=$s timeout -k 10 400 python tools/$t.py > gpurun_out/soak/${t}_$s.log 2>&1 || rc=1
    echo "$t seed $s: $(tail -1 gpurun_out/soak/${t}_$s.log)"
  done
done
exit $rc
