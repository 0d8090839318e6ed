#!/usr/bin/env python3
"""dev tool: randomized differential soak of the PLAIN kernels (pred / fused scan / select) against
numpy: every slot type, ragged sizes, single comparisons, BETWEEN pairs, IN lists, SQL semantics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "20261005")))
TYPES = [(capi.T_INT8, np.int8), (capi.T_INT16, np.int16), (capi.T_INT32, np.int32), (capi.T_INT64, np.int64),
         (capi.T_FLOAT, np.float32), (capi.T_DOUBLE, np.float64)]
OPS = {capi.OP_EQ: np.equal, capi.OP_LT: np.less, capi.OP_LE: np.less_equal, capi.OP_GT: np.greater, capi.OP_GE: np.greater_equal}


def bits_of(words_t, n):
    w = words_t.cpu().numpy().view(np.uint64)
    return np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)


bad = 0
for it in range(300):
    t, npt = TYPES[int(rng.integers(0, len(TYPES)))]
    n = int(rng.choice([1, 31, 1023, 1024, 1025, 2047, 2048, 2049, 4097, int(rng.integers(1, 200000))]))
    span = int(rng.choice([3, 50, 1000]))
    vals = (rng.integers(-span, span, n).astype(npt) if np.issubdtype(npt, np.integer) else rng.normal(0, span, n).astype(npt))
    stride = 8 if npt in (np.int64, np.float64) else 4
    slots = vals.astype({4: np.int32, 8: np.int64}[stride]) if np.issubdtype(npt, np.integer) else vals
    page = torch.from_numpy(np.concatenate([slots.view(np.uint8), np.zeros(16, np.uint8)])).cuda()
    kind = rng.random()
    if kind < 0.5:
        op = int(rng.choice(list(OPS)))
        lit = vals[rng.integers(0, n)]
        exp = OPS[op](vals, lit)
        bm = capi.plain_pred(page, n, t, op, lit)
        res = capi.plain_scan(page, n, t, op, lit)
    elif kind < 0.8:
        lo, hi = np.sort(rng.choice(vals, 2))
        exp = (vals >= lo) & (vals <= hi)
        res = capi.plain_scan(page, n, t, capi.OP_GE, lo, op2=capi.OP_LE, literal2=hi)
        bm = res[0]
    else:
        lst = rng.choice(vals, int(rng.integers(1, 9)))
        exp = np.isin(vals, lst)
        bm = capi.plain_pred(page, n, t, capi.OP_IN, lst)
        res = capi.plain_scan(page, n, t, capi.OP_IN, lst)
    if not np.array_equal(bits_of(bm, n), exp):
        bad += 1
        print("pred mismatch", npt.__name__, n, kind)
        continue
    bitmap, bvals, counts = res
    if not np.array_equal(bits_of(bitmap, n), exp):
        bad += 1
        print("scan bitmap mismatch", npt.__name__, n, kind)
        continue
    c = counts.cpu().numpy()
    bv = bvals.cpu().numpy()
    dense = np.concatenate([bv[b * 2048: b * 2048 + c[b]] for b in range(len(c))]) if len(c) else bv[:0]
    want = slots.view({4: np.int32, 8: np.int64}[stride])[exp]
    if not np.array_equal(dense, want):
        bad += 1
        print("scan values mismatch", npt.__name__, n, kind, len(dense), len(want))
        continue
    sv, sc = capi.plain_select(page, n, t, bitmap)
    scn = sc.cpu().numpy()
    svn = sv.cpu().numpy()
    dense2 = np.concatenate([svn[b * 2048: b * 2048 + scn[b]] for b in range(len(scn))]) if len(scn) else svn[:0]
    if not np.array_equal(dense2, want):
        bad += 1
        print("select mismatch", npt.__name__, n, kind)
print("plain soak done, mismatches:", bad)
sys.exit(1 if bad else 0)
