#!/usr/bin/env python3
"""dev tool: the headline fused scan (w=32, LT @10 %, 2^28 rows), median us; run with
IPS_LIB=<an IPS_ABLATE build> to time it without parts of its materialisation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

capi = entry.load_package().capi
n, bw = 1 << 28, 32
vals = capi.synth_u32(0x5EED0001, n, bw)
enc = capi.fle_encode(vals, bw)
del vals
outs = capi.alloc_scan_outputs(n, torch.device("cuda"))
c = int(0.1 * (1 << 32))
tmin, tmed = timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs), reps=20)
print(f"{os.path.basename(os.environ.get('IPS_LIB', 'libips_hip.so'))}: med {tmed*1e3:.1f} us min {tmin*1e3:.1f} us", flush=True)
