#!/usr/bin/env python3
"""Dev tool: randomized differential soak of the one-pass conjunct chain (ips_chain.hip) -- contiguous
columns through ips_eval_program and page lists with common page ends through ips_eval_program_chunks --
against numpy on the raw values and against the per-operand plan.  IPS_SOAK_SEED, IPS_SOAK_ITERS."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
ips = entry.load_package()
capi = ips.capi
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "1")))
ITERS = int(os.environ.get("IPS_SOAK_ITERS", "60"))
CMP = {O.OP_EQ: np.equal, O.OP_LT: np.less, O.OP_LE: np.less_equal, O.OP_GT: np.greater, O.OP_GE: np.greater_equal}
SIZES = [0, 1, 5, 31, 32, 33, 63, 64, 65, 100, 2047, 2048, 2049, 4097, 10000, 70001]


def dev_words(a):
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return torch.zeros(2, dtype=torch.int64, device="cuda")
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def bits_of(t, n):
    return np.unpackbits(t.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)


bad = 0
for it in range(ITERS):
    n = int(rng.choice([1, 70, 2048, 2049, 3000, 2048 * 7, int(rng.integers(1, int(os.environ.get("IPS_SOAK_MAX_ROWS", "400000"))))]))
    paged = rng.random() < 0.7
    own_cuts = paged and rng.random() < 0.6   # every column cut at its own rows: the segmented chain

    def make_cuts():
        out, left = [], n
        while left > 0:
            s = min(int(rng.choice(SIZES)), left)
            out.append(s)
            left -= s
        return out
    page_rows = make_cuts() if paged else []
    n_ops = int(rng.integers(2, 7))
    cols, keep, nodes, exp = [], [], [], None
    for i in range(n_ops):
        w = int(rng.integers(1, 33))
        span = (1 << w) - 1 if rng.random() < 0.6 else min((1 << w) - 1, 15)
        v = rng.integers(0, span + 1, n, dtype=np.uint64).astype(np.uint32)
        if paged:
            pages, pos = [], 0
            for m in (make_cuts() if own_cuts else page_rows):
                enc = O.fle_encode(v[pos:pos + m], w) if m else np.zeros(2, np.uint64)
                pages.append((dev_words(enc), m, w))
                pos += m
            cols.append(capi.Chunk(pages))
        else:
            keep.append(dev_words(O.fle_encode(v, w)))
            cols.append(capi.fle_column(keep[-1], w))
        kind = rng.integers(0, 3)
        if kind == 0:
            op, c = int(rng.integers(0, 5)), int(rng.integers(0, span + 1))
            nodes.append(capi.leaf(i, op, c))
            sel = CMP[op](v, np.uint32(c))
        elif kind == 1:
            op1, op2 = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            c1, c2 = int(rng.integers(0, span + 1)), int(rng.integers(0, span + 1))
            nodes += [capi.leaf(i, op1, c1), capi.leaf(i, op2, c2)]
            s1, s2 = CMP[op1](v, np.uint32(c1)), CMP[op2](v, np.uint32(c2))
            if rng.random() < 0.7:
                nodes.append(capi.and_node()); sel = s1 & s2
            else:
                nodes.append(capi.or_node()); sel = s1 | s2
        else:
            members = [int(x) for x in rng.integers(0, span + 1, int(rng.integers(1, 17)))]
            nodes.append(capi.leaf(i, O.OP_IN, members))
            sel = np.isin(v, np.array(members, dtype=np.uint32))
        if i == 0:
            exp = sel
        elif rng.random() < 0.75:
            nodes.append(capi.and_node()); exp = exp & sel
        else:
            nodes.append(capi.or_node()); exp = exp | sel
    res = []
    for strat in (capi.PROGRAM_AUTO, capi.PROGRAM_ONE_PASS, capi.PROGRAM_PER_OPERAND):
        capi.set_program_strategy(strat)
        got = capi.eval_program_chunks(nodes, cols) if paged else capi.eval_program(nodes, cols, n)
        res.append(bits_of(got, n))
    capi.set_program_strategy(capi.PROGRAM_AUTO)
    if not all(np.array_equal(r, exp) for r in res):
        bad += 1
        print("chain mismatch: iter", it, "n", n, "paged", paged, "own cuts", own_cuts, "ops", n_ops, [np.array_equal(r, exp) for r in res])
    if paged:
        for c in cols:
            c.close()
print("soak done, mismatches:", bad)
