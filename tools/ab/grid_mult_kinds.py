#!/usr/bin/env python3
"""Dev tool (needs a -DIPS_DEV_KNOBS build, IPS_LIB=...): workgroups per resident slot for the
predicate-only kernels (IPS_GRID_MULT_PRED) and the one-pass chain (IPS_GRID_MULT_CHAIN), every value
timed in the same process, interleaved over three rounds.  Dev builds read the variables on every call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi, q6 = ips.capi, ips.q6
n = 1 << 28
MULTS = (4, 8, 16, 32, 64)
work = {}
keep = []
for w in (4, 6, 8, 12, 16, 24):
    codes = capi.synth_u32(0x5EED0200 + w, n, w)
    enc = capi.fle_encode(codes, w)
    del codes
    keep.append(enc)
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
    c = int(0.3 * (1 << w))
    work[f"fle_pred w={w} LT"] = ("IPS_GRID_MULT_PRED", (lambda enc=enc, w=w, c=c, bm=bm: capi.fle_pred(enc, n, w, 1, [c], bitmap=bm)))
    if w in (4, 12):
        cols = [capi.fle_column(enc, w)]
        nodes = [capi.leaf(0, 4, c // 2), capi.leaf(0, 2, c), capi.and_node()]
        work[f"fle_pred w={w} BETWEEN"] = ("IPS_GRID_MULT_PRED", (lambda nodes=nodes, cols=cols, bm=bm: capi.eval_program(nodes, cols, n, bitmap=bm)))
p32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n + 16,), dtype=torch.int32, device="cuda")
bmp = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
work["plain_pred int32 LT"] = ("IPS_GRID_MULT_PRED", lambda: capi.plain_pred(p32.view(torch.uint8), n, capi.T_INT32, 1, np.int32(-(2 ** 30)), bitmap=bmp))
nq = q6.ROWS
codes = [q6.codes_gpu(capi, c, nq) for c in range(3)]
encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
del codes
nodes, cols = q6.program(capi, encs)
bq = torch.empty((nq + 63) // 64, dtype=torch.int64, device="cuda")


def q6_run(strategy):
    capi.set_program_strategy(strategy)
    capi.eval_program(nodes, cols, nq, bitmap=bq)
    capi.set_program_strategy(capi.PROGRAM_AUTO)


work["Q6 per-operand plan"] = ("IPS_GRID_MULT_PRED", lambda: q6_run(capi.PROGRAM_PER_OPERAND))
work["Q6 one-pass chain"] = ("IPS_GRID_MULT_CHAIN", lambda: q6_run(capi.PROGRAM_ONE_PASS))
for name, (var, fn) in work.items():
    res = {m: [] for m in MULTS}
    for rnd in range(3):
        for m in MULTS:
            os.environ[var] = str(m)
            res[m].append(timeit(fn, reps=10)[1] * 1e3)
    os.environ.pop(var, None)
    print(f"{name:28s} " + "  ".join(f"x{m}: {np.median(res[m]):7.1f}" for m in MULTS), flush=True)
