#!/usr/bin/env python3
"""Dev tool: PLAIN predicate / fused scan for every slot type at 2^28 rows, LT @10 % and BETWEEN."""
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
raw = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)
for name, t, tdt, npt in (("int32", capi.T_INT32, torch.int32, np.int32), ("float", capi.T_FLOAT, torch.float32, np.float32),
                          ("int16", capi.T_INT16, torch.int32, np.int16), ("int64", capi.T_INT64, torch.int64, np.int64),
                          ("double", capi.T_DOUBLE, torch.float64, np.float64)):
    if name == "int32":
        page = raw.clone()
        lit = npt(int(-0.8 * (1 << 31)))
    elif name == "float":
        page = (raw.to(torch.float32) / float(1 << 31)).contiguous()
        lit = npt(-0.8)
    elif name == "int16":
        page = (raw >> 16).contiguous()
        lit = npt(int(-0.8 * (1 << 15)))
    elif name == "int64":
        page = (raw.to(torch.int64) << 8).contiguous()
        lit = npt(int(-0.8 * (1 << 39)))
    else:
        page = (raw.to(torch.float64) / float(1 << 31)).contiguous()
        lit = npt(-0.8)
    stride = 8 if name in ("int64", "double") else 4
    bm = torch.empty(n // 64, dtype=torch.int64, device=raw.device)
    tmin, tmed = timeit(lambda: capi.plain_pred(page, n, t, capi.OP_LT, lit, bitmap=bm))
    sel = capi.bitmap_count(bm, n) / n
    b = n * stride + n // 8
    print(f"plain_pred {name:6s} LT sel={sel:5.3f}  med {tmed * 1e3:7.1f} us  {b / tmed / 1e6:6.0f} GB/s  {b / tmed / 8e9:5.3f}", flush=True)
    tmin, tmed = timeit(lambda: capi.plain_scan(page, n, t, capi.OP_LT, lit), reps=10)
    b2 = b + int(sel * n) * stride
    print(f"plain_scan {name:6s} LT sel={sel:5.3f}  med {tmed * 1e3:7.1f} us  {b2 / tmed / 1e6:6.0f} GB/s  {b2 / tmed / 8e9:5.3f}", flush=True)
    del page, bm
    torch.cuda.empty_cache()
