#!/usr/bin/env python3
"""Dev tool: configs[4] (Q6 shape) on one GPU: ips_eval_program as the one-pass conjunct chain vs
the one-pass chain kernel (ips_set_program_strategy(IPS_PROGRAM_ONE_PASS)), plus two- and four-operand chains."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi, q6 = ips.capi, ips.q6
n = int(os.environ.get("IPS_Q6_ROWS", str(q6.ROWS)))
codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
nodes, cols = q6.program(capi, encs)
bm = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
exp = int(q6.truth(codes).sum().item())
for label, strat in (("per-operand plan", capi.PROGRAM_AUTO), ("one-pass chain", capi.PROGRAM_ONE_PASS)):
    capi.set_program_strategy(strat)
    tmin, tmed = timeit(lambda: capi.eval_program(nodes, cols, n, bitmap=bm), reps=20)
    ok = capi.bitmap_count(bm, n) == exp
    b = q6.algorithmic_bytes(n)
    print(f"Q6 {label:18s} rows={n} min {tmin*1e3:7.1f} us med {tmed*1e3:7.1f} us {b/tmed/1e6:7.0f} GB/s frac {b/tmed/8e9:5.3f} check {ok}", flush=True)
capi.set_program_strategy(capi.PROGRAM_AUTO)
