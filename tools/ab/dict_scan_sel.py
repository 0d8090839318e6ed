#!/usr/bin/env python3
"""dev tool: fused dictionary scan + gather (ips_dict_scan, LE literal) by selectivity and
dictionary size, 2^28 rows: where do the gathers of the selected rows start to cost?"""
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
W = n // 64
for D, bw in ((256, 8), (1024, 10), (4096, 12), (16384, 14), (40000, 16)):
    dv = (np.arange(D, dtype=np.int32) * 5 - 1000)
    codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, bw)
    del codes
    dd = capi.Dict(dv.view(np.uint8), capi.T_INT32)
    outs = None
    line = f"D={D:6d} w={bw:2d}:"
    for sel in (0.01, 0.1, 0.5, 1.0):
        lit = np.array([dv[min(int(sel * D), D - 1)]], dtype=np.int32)
        res = {}
        def f():
            res["r"] = dd.scan(enc, n, bw, capi.OP_LE if sel >= 1.0 else capi.OP_LT, lit)
        tmin, tmed = timeit(f, reps=6)
        nsel = int(res["r"][2].to(torch.int64).sum().item())
        byts = bw * 8 * W + 8 * W + 4 * nsel
        line += f"  @{sel:4.2f} {tmed*1e3:7.1f} us ({byts / tmed / 8e7:4.1f} %)"
        del res
    print(line, flush=True)
    dd.close()
    del enc
