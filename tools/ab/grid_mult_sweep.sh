#!/bin/bash
for rep in 1 2; do
for G in 8 12 16 4; do
  echo "== GRID_MULT $G"
  IPS_GRID_MULT=$G timeout -k 10 300 python tools/kbench.py --bw 32,16,8 --what scan,pred --sel 0.1 2>&1 | grep "scan\|pred"
done
done
