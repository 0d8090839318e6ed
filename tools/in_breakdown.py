#!/usr/bin/env python3
"""dev tool: where the time of the narrow-width IN-list scans goes (2^28 rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    page = np.arange(D, dtype=np.int32) * 3
    dd = capi.Dict(page.view(np.uint8), 2)   # T_INT32
    tmin, tmed = timeit(lambda: dd.decode(enc, n, bw))
    print(f"w={bw:2d}      dict decode (gather every row)      min {tmin*1e3:7.1f} us  med {tmed*1e3:7.1f} us", flush=True)
    for K in (4, 16, 17, 64, 256):
        if K > D:
            continue
        codes = [int(x) for x in np.linspace(1, D - 2, K).astype(int)]
        lits = [c * 3 for c in codes]
        rows = []
        rows.append(("pred IN", timeit(lambda: capi.fle_pred(enc, n, bw, capi.OP_IN, codes, bitmap=outs[0]))))
        rows.append(("scan IN (codes)", timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_IN, codes, outputs=outs))))
        rows.append(("dict scan IN (gather)", timeit(lambda: dd.scan(enc, n, bw, capi.OP_IN, np.array(lits, np.int32)))))
        c_lt = max(int(K * (1 << bw) / D), 1)
        rows.append((f"scan LT {c_lt} (same selectivity)", timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c_lt, outputs=outs))))
        rows.append(("scan EQ (single const)", timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_EQ, codes[0], outputs=outs))))
        for name, (tmin, tmed) in rows:
            print(f"w={bw:2d} K={K:2d} {name:34s} min {tmin*1e3:7.1f} us  med {tmed*1e3:7.1f} us", flush=True)
    dd.close()
