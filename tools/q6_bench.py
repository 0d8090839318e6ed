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

# other chain shapes (2^28 rows, LT at ~30 % per operand): which plans the one-pass kernel wins
n2 = 1 << 28
for widths in ((8, 8), (16, 16), (4, 6, 8, 12), (12, 12, 12, 12), (20, 12), (24, 24), (32, 8), (3, 5)):
    encs2, nodes2, cols2 = [], [], []
    for i, w in enumerate(widths):
        codes2 = capi.synth_u32(0x5EED0100 + i, n2, w)
        encs2.append(capi.fle_encode(codes2, w))
        cols2.append(capi.fle_column(encs2[-1], w))
        nodes2.append(capi.leaf(i, 1, int(0.3 * (1 << w))))
        if i:
            nodes2.append(capi.and_node())
        del codes2
    bm2 = torch.empty((n2 + 63) // 64, dtype=torch.int64, device="cuda")
    res = {}
    for label, strat in (("per-operand", capi.PROGRAM_PER_OPERAND), ("one-pass", capi.PROGRAM_ONE_PASS)):
        capi.set_program_strategy(strat)
        tmin, tmed = timeit(lambda: capi.eval_program(nodes2, cols2, n2, bitmap=bm2), reps=10)
        res[label] = (tmed, capi.bitmap_count(bm2, n2))
    capi.set_program_strategy(capi.PROGRAM_AUTO)
    b = n2 * sum(widths) / 8 + n2 / 8
    print(f"chain {widths}: per-operand {res['per-operand'][0]*1e3:7.1f} us  one-pass {res['one-pass'][0]*1e3:7.1f} us "
          f"({b/res['one-pass'][0]/8e9:5.3f} of 8 TB/s)  counts equal {res['per-operand'][1] == res['one-pass'][1]}", flush=True)
    del encs2, cols2, bm2
