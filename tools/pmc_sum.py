#!/usr/bin/env python3
"""Dev tool: sum rocprofv3 --pmc counters per kernel name from the *_counter_collection.csv files under a directory.
usage: python tools/pmc_sum.py <dir> [name regex]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not pat.search(k):
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
for k, c in acc.items():
    n = len(disp[k])
    print(k[:100], "dispatches", n)
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {v / n:16.1f} per dispatch")
