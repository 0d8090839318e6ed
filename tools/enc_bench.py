#!/usr/bin/env python3
"""dev tool: encoder-side entry points at 2^28 rows (wall time around the synchronous call)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
n = 1 << 28
for D, bw in ((256, 8), (4096, 12), (40000, 16)):
    codes = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32) % D
    vals = (codes.to(torch.int32) * 7 - 1000).contiguous()
    del codes
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        page, w, enc = capi.dict_encode(vals, 2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert w == bw and len(page) == D * 4, (w, len(page))
    print(f"dict_encode int32 D={D:5d}: {dt * 1e3:8.2f} ms  {n / dt / 1e9:7.2f} Grows/s  "
          f"({(4 * n + n * bw / 8) / dt / 1e9:7.1f} GB/s in+out)", flush=True)
    del vals, enc
