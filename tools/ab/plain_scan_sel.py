#!/usr/bin/env python3
"""dev tool: fused PLAIN scan (bitmap + selected slots) by selectivity, int32 and int64, 2^28 rows."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

capi = entry.load_package().capi
n = 1 << 28
for t, npt, width in ((capi.T_INT32, np.int32, 4), (capi.T_INT64, np.int64, 8)):
    x = capi.synth_u32(0x5EED0777, n, 31)
    page = x if width == 4 else x.to(torch.int64)
    page = torch.cat([page.view(torch.uint8), torch.zeros(16, dtype=torch.uint8, device="cuda")])
    del x
    line = f"{npt.__name__}:"
    for sel in (0.01, 0.1, 0.5, 1.0):
        lit = npt(min(int(sel * (1 << 31)), (1 << 31) - 1))
        res = {}
        def f():
            res["r"] = capi.plain_scan(page, n, t, capi.OP_LT if sel < 1.0 else capi.OP_LE, lit)
        tmin, tmed = timeit(f, reps=6)
        nsel = int(res["r"][2].to(torch.int64).sum().item())
        byts = width * n + n // 8 + width * nsel
        line += f"  @{sel:4.2f} {tmed*1e3:7.1f} us ({byts / tmed / 8e7:4.1f} %) sel {nsel / n:.3f}"
        del res
    print(line, flush=True)
    del page
