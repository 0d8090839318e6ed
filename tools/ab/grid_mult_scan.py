#!/usr/bin/env python3
"""Dev tool (needs a -DIPS_DEV_KNOBS build, IPS_LIB=...): the fused scans' workgroups per resident slot
(IPS_GRID_MULT), every value timed in the same process, interleaved over five rounds."""
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
MULTS = (6, 8, 10, 12, 16, 24)
for bw, sel in ((32, 0.1), (32, 0.01), (32, 0.3), (16, 0.1), (12, 0.1), (8, 0.1)):
    vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    enc = capi.fle_encode(vals, bw)
    del vals
    c = int(sel * (1 << bw))
    outs = capi.alloc_scan_outputs(n, torch.device("cuda"))
    fn = lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)
    res = {m: [] for m in MULTS}
    for rnd in range(5):
        for m in MULTS:
            os.environ["IPS_GRID_MULT"] = str(m)
            res[m].append(timeit(fn, reps=10)[1] * 1e3)
    os.environ.pop("IPS_GRID_MULT", None)
    print(f"fle_scan w={bw} LT @{sel:.0%}   " + "  ".join(f"x{m}: {np.median(res[m]):7.1f}" for m in MULTS), flush=True)
    del enc, outs
