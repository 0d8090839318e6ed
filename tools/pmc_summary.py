#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel, average of each counter over the
dispatches with the largest grid (dev tool)."""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    acc = defaultdict(lambda: defaultdict(list))
    grid = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("ips::"):
            continue
        g = int(r["Grid_Size"])
        grid[name] = max(grid.get(name, 0), g)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name in grid and int(r["Grid_Size"]) == grid[name]:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        print(name, "grid", grid[name])
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
