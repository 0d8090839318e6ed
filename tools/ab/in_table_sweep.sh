#!/bin/bash
# list path (IPS_IN_TABLE_MIN=1000) against the membership-table path (=1); no variable: the defaults
for M in "" 1000 1; do echo "== IPS_IN_TABLE_MIN=$M"; IPS_IN_TABLE_MIN=$M timeout -k 10 500 python tools/ab/in_table_sweep.py 2>&1 | grep "w="; done
