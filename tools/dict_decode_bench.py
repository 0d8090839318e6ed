#!/usr/bin/env python3
"""dev tool: ips_dict_decode (gather every row) for several dictionary sizes, int32 and int64."""
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
for bw in (8, 10, 12, 13, 14, 16):
    D = 40000 if bw == 16 else 1 << bw
    codes = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32) % D
    enc = capi.fle_encode(codes.to(torch.int32), bw)
    del codes
    for tname, t, npt in (("int32", 2, np.int32), ("int64", 3, np.int64)):
        page = (np.arange(D, dtype=npt) * 3 - 7)
        dd = capi.Dict(page.view(np.uint8), t)
        out = torch.empty(n, dtype=capi.TORCH_SLOT[t], device="cuda")
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        import ctypes as C
        f = lambda: capi.lib().ips_dict_decode(dd.h, C.c_void_p(enc.data_ptr()), C.c_int64(n), bw,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(bad.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
        tmin, tmed = timeit(f, reps=10)
        byts = n * bw / 8 + n * page.itemsize
        print(f"w={bw:2d} D={D:6d} {tname}: med {tmed*1e3:7.1f} us  {byts / tmed / 1e6:7.0f} GB/s", flush=True)
        dd.close()
        del out
