#!/usr/bin/env python3
"""dev tool: late materialisation of an OPTIONAL dictionary column (ips_dict_select_nullable) on
2^28 rows, 10 % NULL, by selectivity: us per call and the share of each launch (run under
rocprofv3 --kernel-trace --stats for the latter)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

capi = entry.load_package().capi
n, bw, D = 1 << 28, 12, 4096
dev = torch.device("cuda")
nn = capi.synth_u32(0x5EED0D1, n, 32)
is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
defs = capi.fle_encode(is_set.to(torch.int32), 1)
k = int(is_set.sum().item())
del nn, is_set
codes = (capi.synth_u32(0x5EED0D2, k, 32).to(torch.int64) & 0xFFFFFFFF) % D
enc = capi.fle_encode(codes.to(torch.int32), bw)
del codes
dd = None if os.environ.get("IPS_SELNULL_RAW") else capi.Dict((np.arange(D, dtype=np.int32) * 3 + 7).view(np.uint8), capi.T_INT32)
for sel in (0.01, 0.1, 0.5):
    s = capi.synth_u32(0x5EED0D3, n, 32)
    bm = capi.bitmap_from_bool((s.to(torch.int64) & 0xFFFFFFFF) < int(sel * (1 << 32))) if hasattr(capi, "bitmap_from_bool") else None
    if bm is None:
        senc = capi.fle_encode(((s.to(torch.int64) & 0xFFFFFFFF) >> 16).to(torch.int32), 16)
        bm = capi.fle_pred(senc, n, 16, capi.OP_LT, int(sel * 65536))
        del senc
    del s
    nsel = capi.bitmap_count(bm, n)
    tmin, tmed = timeit(lambda: capi.select_nullable(dd, defs, 1, 1, n, enc, k, bw, bm), reps=8)
    alg = n / 8 * 2 + k * bw / 8 + nsel * 4 + nsel / 8
    print(f"select_nullable sel={sel}: selected {nsel}  med {tmed*1e3:8.1f} us  min {tmin*1e3:8.1f} us  algorithmic {alg/1e6:.0f} MB -> {alg/tmed/1e6:.0f} GB/s ({alg/tmed/8e7:.1f} %)", flush=True)
