#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for G in 4 8; do echo "== GRID_MULT $G"; IPS_GRID_MULT=$G timeout -k 10 300 python tools/q6_bench.py 2>&1 | grep per-operand; IPS_GRID_MULT=$G timeout -k 10 300 python tools/nullable_bench.py --bw 12 2>&1 | grep "nullable leaf"; IPS_GRID_MULT=$G timeout -k 10 300 python tools/configs_bench.py 2>&1 | grep "PLAIN int64 BETWEEN sel=0.1\|D=4096 w=12 IN K=16\|D=256 w=8 IN K=16" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:80], d['us_med'])"; done; done
