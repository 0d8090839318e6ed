#!/usr/bin/env python3
"""Dev tool: IN-list scans + gather at 2^28 rows, list path vs membership-table path
(IPS_IN_TABLE_MIN=1000 forces the list path up to 256 constants, =1 the table path)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi = ips.capi
n = 1 << 28
dev = torch.device("cuda")
for bw in (8, 12, 16):
    vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    enc = capi.fle_encode(vals, bw)
    outs = capi.alloc_scan_outputs(n, dev)
    D = 1 << bw
    for K in (8, 12, 16, 20, 24, 32, 48, 64, 128):
        if K > D // 4:
            continue
        codes = [int(x) for x in np.linspace(1, D - 2, K).astype(int)]
        tmin, tmed = timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_IN, codes, outputs=outs))
        tmin2, tmed2 = timeit(lambda: capi.fle_pred(enc, n, bw, capi.OP_IN, codes, bitmap=outs[0]))
        print(f"w={bw:2d} K={K:3d} scan med {tmed*1e3:7.1f} us   pred med {tmed2*1e3:7.1f} us", flush=True)
