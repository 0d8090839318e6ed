#!/usr/bin/env python3
"""Kernel resource usage (VGPRs, SGPRs, spills, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... --cuda-device-only -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2> r.txt
       python tools/kres.py r.txt '<regex on the demangled name>'"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
rows = []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    rows.append((name, g("VGPRs"), g("TotalSGPRs"), g("VGPRs Spill"), g("SGPRs Spill"),
                 g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"ScratchSize \[bytes/lane\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for n, r in zip(names, rows):
    n = n.split("(")[0]
    if re.search(pat, n):
        print(f"{n[:80]:80s} vgpr {r[1]:>3s} sgpr {r[2]:>3s} vspill {r[3]:>3s} sspill {r[4]:>3s} lds {r[5]:>6s} occ {r[6]} scratch {r[7]}")
