#!/usr/bin/env python3
"""dev tool: ips_fle_pred at w=32 with the early-pruning kernel on friendly and hostile columns."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi = ips.capi
n = 1 << 28
full = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)
cases = [("uniform 32-bit values, LT 10 %", full, int(0.1 * (1 << 32))),
         ("values < 2^16 (high planes all equal the constant's), LT 6554", full & 0xFFFF, 6554),
         ("values < 2^16, LT 2^31 (decided by the first plane)", full & 0xFFFF, 1 << 31),
         ("every other 2048-row tile < 2^16, LT 6554",
          torch.where(((torch.arange(n, device="cuda") >> 11) & 1) == 0, full & 0xFFFF, full), 6554)]
bm = torch.empty(n // 64 + 2, dtype=torch.int64, device="cuda")
for name, vals, c in cases:
    enc = capi.fle_encode(vals.to(torch.int32), 32)
    tmin, tmed = timeit(lambda: capi.fle_pred(enc, n, 32, capi.OP_LT, c, bitmap=bm), reps=20)
    cnt = capi.bitmap_count(bm, n)
    exp = int(((vals.to(torch.int64) & 0xFFFFFFFF) < c).sum().item())   # the column is unsigned
    print(f"{name:62s} med {tmed * 1e3:7.1f} us  ok={cnt == exp}", flush=True)
    del enc
